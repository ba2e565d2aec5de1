"""Ad-hoc probe (not a test): per-block (start, end, CU) trace of every k_dt_pass launch of one person-model frame.

    python tests/tools_dt_trace.py [W H [launch [B]]]     B > 1: a batch of B frames (the first 4096 blocks of every launch are traced;
                                                          the phase table then covers blocks 1536.., which start on a loaded chip)
"""
import os
os.environ.setdefault("PBD_LIBRARY", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "partsbaseddetector_amd", "libpbd_hip_probes.so"))  # `make -C partsbaseddetector_amd/csrc probes`
os.environ["PBD_DT_TRACE"] = "1"
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from partsbaseddetector_amd import capi
from partsbaseddetector_amd.model import make_image, make_person_model

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 480)
NBATCH = int(sys.argv[4]) if len(sys.argv) > 4 else 1
model = make_person_model(K=6)
model.thresh = 1e9
h = capi.Handle(model, conv_mode=capi.PBD_CONV_MFMA, graph=0)
img = torch.from_numpy(make_image(0, W, H)).cuda()
frames = [make_image(i, W, H) for i in range(NBATCH)]
def run():
    if NBATCH > 1:
        h.detect_batch(frames, capacity=16)
    else:
        h.enqueue_dev(img.data_ptr(), W, H, 3); h.collect(16)
L = capi.lib()
NL, NB = 40, 4096
t = np.zeros((NL, NB, 8), np.uint64)
hw = np.zeros((NL, NB), np.uint32)
nl = C.c_int(0)
def read():
    L.pbd_debug_dt_trace(t.ctypes.data_as(C.POINTER(C.c_ulonglong)), hw.ctypes.data_as(C.POINTER(C.c_uint)), C.byref(nl))
for i in range(3):
    run()
read()                       # resets the launch counter
t[:] = 0
_st = (C.c_ulonglong * 8)(); L.pbd_debug_dt_stamps(_st)     # resets the redo counter
run()
read()
st = (C.c_ulonglong * 8)()
L.pbd_debug_dt_stamps(st)
print("launches traced:", nl.value, "| lines redone sequentially since the last read:", st[7], "| PBD_DT_BF_MAXLEN", os.environ.get("PBD_DT_BF_MAXLEN"))
for l in range(min(nl.value, NL)):
    s, e = t[l, :, 0].astype(np.int64), t[l, :, 7].astype(np.int64)
    nb = int((s > 0).sum())
    if nb == 0: continue
    s, e = s[:nb], e[:nb]
    t0 = s.min()
    st, en, du = (s - t0) / 100.0, (e - t0) / 100.0, (e - s) / 100.0
    last = int(en.argmax())
    cu = hw[l, :nb] & 0xffffff
    xcc = hw[l, :nb] >> 24
    # HW_ID: wave 3:0 simd 5:4 pipe 7:6 cu 11:8 sh 12 se 15:13
    cuid = ((hw[l, :nb] >> 8) & 0xff) | (xcc << 8)
    print(f"launch {l:2d}: blocks {nb:5d} span {en.max():6.1f} us | started by: 50% {np.percentile(st, 50):5.1f} 99% {np.percentile(st, 99):5.1f} max {st.max():5.1f} | "
          f"dur p50 {np.percentile(du, 50):5.1f} p90 {np.percentile(du, 90):5.1f} max {du.max():5.1f} (blk {int(du.argmax())}) | last end: blk {last} start {st[last]:5.1f} dur {du[last]:5.1f} | "
          f"CUs {len(np.unique(cuid))} first-32 mean dur {du[:32].mean():5.1f} | started < 2 us: {int((st < 2).sum())}, dur > 1.5 x p50: {int((du > 1.5 * np.percentile(du, 50)).sum())}")
if len(sys.argv) > 3:
    l = int(sys.argv[3])
    s, e = t[l, :, 0].astype(np.int64), t[l, :, 7].astype(np.int64)
    nb = int((s > 0).sum()); t0 = s[:nb].min()
    du = (e[:nb] - s[:nb]) / 100.0
    ph = t[l, :nb, :].astype(np.int64)
    names = ["setup", "load", "scan", "stitch", "validate", "readout"]
    order = [0, 1, 2, 3, 6, 4, 5]          # stamp indices in program order
    seg = np.stack([(ph[:, order[i + 1]] - ph[:, order[i]]) / 100.0 for i in range(6)], 1)
    BS = 64 if NBATCH == 1 else 512
    if NBATCH > 1:
        lo = min(1536, nb // 2)
        print(f"batch of {NBATCH}: blocks {lo}..{nb - 1} (started on a loaded chip): mean dur %.1f |" % du[lo:].mean(),
              " ".join(f"{n} {seg[lo:, i].mean():.2f}" for i, n in enumerate(names)), "| blocks 0..%d: mean dur %.1f |" % (lo - 1, du[:lo].mean()),
              " ".join(f"{n} {seg[:lo, i].mean():.2f}" for i, n in enumerate(names)))
    for b in range(0, nb, 64 if NBATCH == 1 else 512):
        print(b, "mean dur %.1f max %.1f |" % (du[b:b + BS].mean(), du[b:b + BS].max()), " ".join(f"{n} {seg[b:b + BS, i].mean():.1f}/{seg[b:b + BS, i].max():.1f}" for i, n in enumerate(names)))
    worst = np.argsort(-du)[:12]
    for b in worst:
        print("slow blk", int(b), "dur %.1f |" % du[b], " ".join(f"{n} {seg[b, i]:.1f}" for i, n in enumerate(names)))
h.close()
