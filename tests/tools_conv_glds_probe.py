"""Ad-hoc probe (not a test): per-phase time sums of one workgroup of the persistent filter bank k_conv_glds (probe build)."""
import os
os.environ.setdefault("PBD_LIBRARY", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "partsbaseddetector_amd", "libpbd_hip_probes.so"))
os.environ.setdefault("PBD_MFMA_VARIANT", "10")
import ctypes as C
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from partsbaseddetector_amd import capi
from partsbaseddetector_amd.model import make_image, make_person_model
m = make_person_model(); m.thresh = 3e38
h = capi.Handle(m, conv_mode=capi.PBD_CONV_MFMA)
im = make_image(0, 640, 480)
for _ in range(3):
    h.detect(im)
    st = (C.c_ulonglong * 8)()
    capi.lib().pbd_debug_conv_stamps(st)
    n = max(1, st[6])
    kus = st[1] / 100.0
    print(f"variant {os.environ['PBD_MFMA_VARIANT']}: units {st[6]}, us per unit: barrier waits {st[0] / 100.0 / n:.2f} K loops {kus / n:.2f} "
          f"barrier {st[2] / 100.0 / n:.2f} epilogue {st[5] / 100.0 / n:.2f} | life {st[7] / 100.0:.1f} us | shader clock in the K loops "
          f"{st[4] / max(kus, 1e-9) / 1e3:.3f} GHz, {kus / n / 1.6:.2f} ns per MFMA of this wave (1600 per unit when all M-tiles are valid)", flush=True)
h.close()
