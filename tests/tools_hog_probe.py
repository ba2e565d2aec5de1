"""Ad-hoc probe (not a test): per-phase timestamps of block 0 of k_hog."""
import os
os.environ.setdefault("PBD_LIBRARY", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "partsbaseddetector_amd", "libpbd_hip_probes.so"))  # `make -C partsbaseddetector_amd/csrc probes`
import ctypes as C
import sys
sys.path.insert(0, "/root/repo")
from partsbaseddetector_amd import capi
from partsbaseddetector_amd.model import make_image, make_tree_model
h = capi.Handle(make_tree_model([-1, 0], 1, seed=1), conv_mode=capi.PBD_CONV_EXACT)
for _ in range(3):
    h.hog(make_image(0, 640, 480))
    st = (C.c_ulonglong * 8)()
    capi.lib().pbd_debug_hog_stamps(st)
    d = [(st[i + 1] - st[i]) / 100.0 for i in range(5)]
    print(f"hog block0 phases us: stage {d[0]:.1f} gradient {d[1]:.1f} histogram {d[2]:.1f} energy+norm {d[3]:.1f} features {d[4]:.1f}", flush=True)
h.close()
