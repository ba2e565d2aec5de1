"""Ad-hoc probe (not a test): saturated throughput of each stage alone — S host threads, one handle each, looping ONE stage
(the stage entry points synchronise, so S threads keep S streams busy).  Tells which stage the 4-frames-in-flight
throughput is spent in: ms per frame per stage at saturation vs the sequential stage times."""
import os
import sys
import threading
import time
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from partsbaseddetector_amd import capi
from partsbaseddetector_amd.model import make_image, make_person_model

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dp_mode = int(os.environ.get("DP_MODE", "1"))
W, H = 640, 480
model = make_person_model(K=6)
model.thresh = 1e9
hs = [capi.Handle(model, conv_mode=capi.PBD_CONV_MFMA, graph=0, dp_mode=dp_mode) for _ in range(S)]
im = make_image(0, W, H)
for h in hs:
    h.pyramid(im); h.pdf(); h.dp_min()

def loop(h, fn, n):
    for _ in range(n):
        fn(h)

def sat(name, fn):
    ths = [threading.Thread(target=loop, args=(h, fn, N)) for h in hs]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name}: {dt / (N * S) * 1e3:.4f} ms per frame at saturation ({S} streams)", flush=True)

sat("pyramid+hog", lambda h: h.pyramid(im))
sat("pdf", lambda h: h.pdf())
sat("dp_min", lambda h: h.dp_min())
def all3(h):
    h.pyramid(im); h.pdf(); h.dp_min()
sat("pyramid+hog+pdf+dp (stage calls)", all3)
for h in hs: h.close()
