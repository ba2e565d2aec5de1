"""Ad-hoc probe (not a test, CPU only): dynamic statistics of the segment-parallel distance transform on REAL response lines — the
oracle's filter responses of one 640x480 person frame, rows (x pass) and columns (y pass) of 12 filters at levels 0 / 5 / 12 / 20 —
for the planner's two float block geometries (128 lanes / 25 KB and 256 lanes / 40 KB): scan steps per segment and stitch
iterations per boundary, mean and per-wavefront maximum (tests/tools/dt_line_stats.cpp compiles dt_core.hpp for the host).

    python tests/tools_dt_line_stats.py > profiles/r04_dt_line_stats.txt
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402
from partsbaseddetector_amd.model import make_image, make_person_model  # noqa: E402

m = make_person_model(K=6)
m.thresh = 1e30
fr = orc.detect(m, make_image(0, 640, 480), capacity=1, keep=True)[4]
with tempfile.TemporaryDirectory() as td:
    with open(os.path.join(td, "lines.bin"), "wb") as f:
        for l in (0, 5, 12, 20):
            if l >= fr.nlevels:
                continue
            r = np.asarray(fr.resp(l))[:12]                    # [12 filters, H, W]
            for lines in (r.reshape(-1, r.shape[2]), np.ascontiguousarray(r.transpose(0, 2, 1)).reshape(-1, r.shape[1])):
                np.asarray(lines.shape, np.int32).tofile(f)
                np.ascontiguousarray(lines, np.float32).tofile(f)
    fr.free()
    exe = os.path.join(td, "stat")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "partsbaseddetector_amd", "csrc"),
                           os.path.join(ROOT, "tests", "tools", "dt_line_stats.cpp"), "-o", exe])
    print("# Dynamic statistics of the segment-parallel distance transform on REAL response lines (person 26 x 6 model, 640x480 seed frame,")
    print("# levels 0 / 5 / 12 / 20, x-pass rows and y-pass columns of 12 filters each; a = -0.02, b = 0.003), block geometry as the planner's.")
    print('# "wave-max" = mean over groups of 64 consecutive segments / boundaries of the group\'s maximum (what a wavefront executes).')
    for kb, nt in ((25, 128), (40, 256)):
        print(f"## {nt} lanes per block, {kb} KB of LDS")
        sys.stdout.flush()
        subprocess.check_call([exe, str(kb), str(nt)], cwd=td)
