#!/bin/bash
# r05 session 19: pbd_tune_plan (test), then the tuner on the benched shape and on two other sizes (single frames and batches of 8)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s19; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x -s -k "tune_plan" > $O/pytest_tune.log 2>&1; echo "rc=$?" >> $O/pytest_tune.log; grep -E "tune_plan|passed|failed|Error" $O/pytest_tune.log | tail -8
timeout 600 python - > $O/tune_sizes.log 2>&1 <<'PY'
import numpy as np
from partsbaseddetector_amd import capi
from partsbaseddetector_amd.model import make_image, make_person_model
m = make_person_model(K=6)
m.thresh = 3e38
for (w, h) in ((640, 480), (480, 360), (800, 600), (1280, 720)):
    hd = capi.Handle(m)
    im = make_image(0, w, h)
    for batch in (1, 8):
        if w * h > 800 * 600 and batch > 1:
            continue
        c, ms = hd.tune_plan(im, batch=batch)
        print(f"{w}x{h} batch {batch}: chosen {c} (1 = 256 lanes / 40 KB, 2 = 128 lanes / 25 KB)  dp_min ms per call {ms[0]:.4f} / {ms[1]:.4f}  per frame {ms[0]/batch:.4f} / {ms[1]/batch:.4f}", flush=True)
    hd.close()
PY
echo "rc=$?"; cat $O/tune_sizes.log | tail -12
