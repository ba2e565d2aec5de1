"""Ad-hoc probe (not a test): per-phase timestamps of one workgroup of k_conv_mfma on the person model."""
import os
os.environ.setdefault("PBD_LIBRARY", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "partsbaseddetector_amd", "libpbd_hip_probes.so"))  # `make -C partsbaseddetector_amd/csrc probes`
import ctypes as C
import sys
sys.path.insert(0, "/root/repo")
from partsbaseddetector_amd import capi
from partsbaseddetector_amd.model import make_image, make_person_model
m = make_person_model(); m.thresh = 3e38
h = capi.Handle(m, conv_mode=capi.PBD_CONV_MFMA)
im = make_image(0, 640, 480)
for _ in range(3):
    h.detect(im)
    st = (C.c_ulonglong * 8)()
    capi.lib().pbd_debug_conv_stamps(st)
    d = [(st[i + 1] - st[i]) / 100.0 for i in range(6)]   # k_conv_mfma16 stamps: 0 start, 1/3 staged half 0/1, 2/4 K loop of half 0/1 done, 5 barrier, 6 end
    print(f"conv WG(300,2) phases us: stage0 {d[0]:.1f} kloop0 {d[1]:.1f} stage1 {d[2]:.1f} kloop1 {d[3]:.1f} barrier {d[4]:.1f} epilogue {d[5]:.1f} total {(st[6] - st[0]) / 100.0:.1f}", flush=True)
h.close()
