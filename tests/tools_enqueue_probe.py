"""Ad-hoc probe (not a test): host-side cost of one pbd_detect_enqueue_dev_u8 + collect (CPU time per frame)."""
import os
os.environ.setdefault("PBD_LIBRARY", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "partsbaseddetector_amd", "libpbd_hip_probes.so"))  # `make -C partsbaseddetector_amd/csrc probes`
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from partsbaseddetector_amd import capi
from partsbaseddetector_amd.model import make_image, make_person_model
m = make_person_model(K=6); m.thresh = 3e38
im = torch.from_numpy(make_image(0, 640, 480)).cuda()
h = capi.Handle(m)
for _ in range(5): h.detect_dev(im.data_ptr(), 640, 480, 3)
torch.cuda.synchronize()
te = tc = 0.0
N = 200
for _ in range(N):
    t0 = time.perf_counter(); h.enqueue_dev(im.data_ptr(), 640, 480, 3); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    h.collect(); t3 = time.perf_counter()
    te += t1 - t0; tc += t3 - t2
print(f"enqueue (host) {te / N * 1e6:.1f} us/frame, collect after completion {tc / N * 1e6:.1f} us/frame")
