import os
os.environ.setdefault("PBD_LIBRARY", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "partsbaseddetector_amd", "libpbd_hip_probes.so"))  # `make -C partsbaseddetector_amd/csrc probes`
import sys, ctypes as C
import numpy as np
sys.path.insert(0, "/root/repo")
from partsbaseddetector_amd import capi
from partsbaseddetector_amd.model import make_tree_model
from oracle import orc
for cp in (1, 0):
    h = capi.Handle(make_tree_model([-1, 0], 1, seed=1), conv_mode=capi.PBD_CONV_EXACT, dt_mode=2, dt_correct_ptr=cp)
    rng = np.random.default_rng(7)
    for (r, c) in [(2, 130), (4, 130), (4, 60), (4, 4), (130, 4), (130, 2), (20, 70)]:
        a = rng.normal(0, 1.5, (r, c)).astype(np.float32)
        got = h.dt2d(a, -0.02, 0.003, -0.03, 0.001, 0, 0)
        ref = orc.dt2d(a, -0.02, 0.003, -0.03, 0.001, 0, 0, correct_ptr=cp)
        bad = [int((got[k] != ref[k]).sum()) for k in range(3)]
        print("cp", cp, (r, c), "mismatches score/ix/iy", bad)
        for k in (1, 2):
            if bad[k]:
                ys, xs = np.nonzero(got[k] != ref[k])
                print("   k", k, "at", list(zip(ys[:8], xs[:8])), "got", got[k][ys[:8], xs[:8]], "ref", ref[k][ys[:8], xs[:8]])
    h.close()
