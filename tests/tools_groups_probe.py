import os
os.environ.setdefault("PBD_LIBRARY", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "partsbaseddetector_amd", "libpbd_hip_probes.so"))  # `make -C partsbaseddetector_amd/csrc probes`
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from partsbaseddetector_amd import capi
from partsbaseddetector_amd.model import make_image, make_person_model
m = make_person_model(K=6); m.thresh = 3e38
im = torch.from_numpy(make_image(0, 640, 480)).cuda()
for g in (1, 2, 3):
    h = capi.Handle(m, dp_groups=g)
    h.set_profiling(True)
    for _ in range(5): h.detect_dev(im.data_ptr(), 640, 480, 3)
    acc = {}
    for _ in range(20):
        h.detect_dev(im.data_ptr(), 640, 480, 3)
        for k, v in h.stage_ms().items(): acc[k] = acc.get(k, 0) + v / 20
    print("groups", g, {k: round(v, 4) for k, v in acc.items()})
    h.close()
