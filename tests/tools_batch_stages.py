"""Ad-hoc probe (not a test): per-stage GPU time of a BATCH of frames on one handle (HIP events, eager launches)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from partsbaseddetector_amd import capi
from partsbaseddetector_amd.model import make_image, make_person_model

W, H = 640, 480
model = make_person_model(K=6)
model.thresh = 1e9
for B in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    h = capi.Handle(model, graph=0, max_candidates=4096)
    dev = torch.from_numpy(np.stack([make_image(i, W, H) for i in range(B)])).cuda()
    for _ in range(3):
        h.enqueue_batch_dev(dev.data_ptr(), B, W, H, 3); h.collect_batch(16)
    h.set_profiling(True)
    acc = {}
    N = 10
    for _ in range(N):
        h.enqueue_batch_dev(dev.data_ptr(), B, W, H, 3); h.collect_batch(16)
        for k, v in h.stage_ms().items():
            acc[k] = acc.get(k, 0.0) + v / N / B
    print(f"batch {B}: ms per FRAME: " + " ".join(f"{k} {v:.4f}" for k, v in acc.items()), flush=True)
    h.close()
