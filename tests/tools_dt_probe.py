"""Ad-hoc probe (not a test): standalone DT on a few map sizes, per-phase timestamps of block 0."""
import os
os.environ.setdefault("PBD_LIBRARY", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "partsbaseddetector_amd", "libpbd_hip_probes.so"))  # `make -C partsbaseddetector_amd/csrc probes`
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, "/root/repo")
from partsbaseddetector_amd import capi
from partsbaseddetector_amd.model import make_tree_model

h = capi.Handle(make_tree_model([-1, 0], 1, seed=1), conv_mode=capi.PBD_CONV_EXACT)
rng = np.random.default_rng(0)
L = capi.lib()
for (r, c) in [(8, 10), (118, 158), (118, 158), (16, 158), (158, 16), (64, 64)]:
    a = rng.normal(0, 1.5, (r, c)).astype(np.float32)
    h.dt2d(a, -0.01, 0.001, -0.02, -0.002, 1, -1)
    st = (C.c_ulonglong * 8)()
    L.pbd_debug_dt_stamps(st)
    us = lambda a, b: (st[b] - st[a]) / 100.0
    print(f"{r}x{c} y-pass block0 phases us: setup+rtable {us(0, 1):.1f} load {us(1, 2):.1f} segment scans {us(2, 3):.1f} "
          f"stitches {us(3, 6):.1f} validate+table {us(6, 4):.1f} read-out {us(4, 5):.1f}; lines redone sequentially: {st[7]}", flush=True)
h.close()
