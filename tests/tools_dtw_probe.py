"""Ad-hoc probe (not a test): wave-per-line DT on single maps under the kernel tracer."""
import os
os.environ.setdefault("PBD_LIBRARY", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "partsbaseddetector_amd", "libpbd_hip_probes.so"))  # `make -C partsbaseddetector_amd/csrc probes`
import sys
import numpy as np
sys.path.insert(0, "/root/repo")
from partsbaseddetector_amd import capi
from partsbaseddetector_amd.model import make_tree_model
h = capi.Handle(make_tree_model([-1, 0], 1, seed=1), conv_mode=capi.PBD_CONV_EXACT, dt_mode=2)
rng = np.random.default_rng(0)
for (r, c) in [(16, 158), (16, 158), (16, 64), (16, 32), (64, 158), (118, 158)]:
    a = rng.normal(0, 1.5, (r, c)).astype(np.float32)
    h.dt2d(a, -0.01, 0.001, -0.02, -0.002, 1, -1)
h.close()
