"""Ad-hoc probe (not a test): how the blocks of every k_dt_pass launch of ONE person-model frame spread over the CUs (probe build's block trace):
blocks per CU histogram, and the blocks' durations by how many blocks shared their CU.

    python tests/tools_dt_cu_spread.py [W H]
"""
import os
os.environ.setdefault("PBD_LIBRARY", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "partsbaseddetector_amd", "libpbd_hip_probes.so"))
os.environ["PBD_DT_TRACE"] = "1"
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from partsbaseddetector_amd import capi
from partsbaseddetector_amd.model import make_image, make_person_model

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 480)
model = make_person_model(K=6)
model.thresh = 1e9
h = capi.Handle(model, conv_mode=capi.PBD_CONV_MFMA, graph=0)
img = torch.from_numpy(make_image(0, W, H)).cuda()
L = capi.lib()
NL, NB = 40, 4096
t = np.zeros((NL, NB, 8), np.uint64)
hw = np.zeros((NL, NB), np.uint32)
nl = C.c_int(0)
def run():
    h.enqueue_dev(img.data_ptr(), W, H, 3); h.collect(16)
def read():
    L.pbd_debug_dt_trace(t.ctypes.data_as(C.POINTER(C.c_ulonglong)), hw.ctypes.data_as(C.POINTER(C.c_uint)), C.byref(nl))
for i in range(3):
    run()
read(); t[:] = 0
run(); read()
for l in range(min(nl.value, NL)):
    s, e = t[l, :, 0].astype(np.int64), t[l, :, 7].astype(np.int64)
    nb = int((s > 0).sum())
    if nb == 0:
        continue
    du = (e[:nb] - s[:nb]) / 100.0
    x = hw[l, :nb]
    # HW_ID: wave 3:0 simd 5:4 pipe 7:6 cu 11:8 sh 12 se 15:13; xcc in bits 24+
    cu = ((x >> 8) & 0xff) | ((x >> 24) << 8)
    ids, cnt = np.unique(cu, return_counts=True)
    per = dict(zip(ids.tolist(), cnt.tolist()))
    share = np.array([per[int(c)] for c in cu])
    hist = np.bincount(cnt, minlength=9)[1:9]
    by = " ".join(f"{k}/CU: {du[share == k].mean():.1f}us(n={int((share == k).sum())})" for k in range(1, 9) if (share == k).any())
    print(f"launch {l:2d}: {nb:5d} blocks on {len(ids)} CUs | CUs holding 1..8 blocks: {hist.tolist()} | span {(e[:nb].max() - s[:nb].min()) / 100.0:.1f} us | mean dur by blocks sharing the CU: {by}")
