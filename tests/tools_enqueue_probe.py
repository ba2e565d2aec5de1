"""Ad-hoc probe (not a test): host time to enqueue one frame and to collect it, eager launches vs hipGraph replay,
and the sequential frame wall time minus the sum of its kernels' GPU time."""
import os
os.environ.setdefault("PBD_LIBRARY", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "partsbaseddetector_amd", "libpbd_hip_probes.so"))  # `make -C partsbaseddetector_amd/csrc probes`
import sys
import time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from partsbaseddetector_amd import capi
from partsbaseddetector_amd.model import make_image, make_person_model

m = make_person_model(); m.thresh = 3e38
im = torch.from_numpy(make_image(0, 640, 480)).cuda()
torch.cuda.synchronize()
for graph in (0, 1):
    h = capi.Handle(m, graph=graph)
    for _ in range(20):
        h.detect_dev(im.data_ptr(), 640, 480, 3)
    te, tc, tw = [], [], []
    for _ in range(200):
        t0 = time.perf_counter(); h.enqueue_dev(im.data_ptr(), 640, 480, 3); t1 = time.perf_counter()
        h.collect(); t2 = time.perf_counter()
        te.append(t1 - t0); tw.append(t2 - t0)
    print(f"graph={graph}: enqueue {np.median(te) * 1e6:.1f} us (p90 {np.percentile(te, 90) * 1e6:.1f}), frame wall {np.median(tw) * 1e3:.4f} ms", flush=True)
    h.close()
