"""Ad-hoc probe (not a test): do the filter bank (MFMA) and the DP / distance transform (vector ALU, LDS) overlap on the chip?
P host threads loop pdf() and D threads loop dp_min() on their own handles for a fixed time; rates alone vs together."""
import os
import sys
import threading
import time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from partsbaseddetector_amd import capi
from partsbaseddetector_amd.model import make_image, make_person_model

T = float(sys.argv[1]) if len(sys.argv) > 1 else 0.4
W, H = 640, 480
model = make_person_model(K=6)
model.thresh = 1e9
NH = 6
hs = [capi.Handle(model, conv_mode=capi.PBD_CONV_MFMA, graph=0) for _ in range(NH)]
im = make_image(0, W, H)
for h in hs:
    h.pyramid(im); h.pdf(); h.dp_min()

def run(np_, nd):
    stop = [False]
    counts = [0] * (np_ + nd)
    def loop(i, h, fn):
        while not stop[0]:
            fn(h); counts[i] += 1
    ths = [threading.Thread(target=loop, args=(i, hs[i], (lambda h: h.pdf()) if i < np_ else (lambda h: h.dp_min()))) for i in range(np_ + nd)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in ths: t.start()
    time.sleep(T); stop[0] = True
    for t in ths: t.join()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    rp = sum(counts[:np_]) / dt; rd = sum(counts[np_:]) / dt
    print(f"{np_} pdf + {nd} dp_min streams: pdf {rp:8.1f} /s ({1e3 / rp if rp else 0:.3f} ms each)  dp_min {rd:8.1f} /s ({1e3 / rd if rd else 0:.3f} ms each)", flush=True)
    return rp, rd

for cfg in [(2, 0), (3, 0), (0, 2), (0, 3), (2, 2), (3, 3), (1, 3), (2, 4)]:
    run(*cfg)
for h in hs: h.close()
