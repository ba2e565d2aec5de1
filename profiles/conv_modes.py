#!/usr/bin/env python
"""pdf() on the GPU: direct VALU correlation (PBD_CONV_EXACT, reference summation order) vs fp32 MFMA implicit GEMM
(PBD_CONV_MFMA) vs the split-product bank on the bf16 matrix units (PBD_CONV_SPLIT, round 5) for the person model with K = 1, 2, 6, 8, 12
mixtures per part (N = 26, 52, 156, 208, 312 filters) at 640x480 — the measurement behind PBD_CONV_AUTO's rule (BASELINE configs[4]: "MFMA im2col-GEMM vs direct conv,
rocprof-chosen").  Stage times are HIP events around the filter-bank launch (pbd_get_stage_ms); run under
`rocprofv3 --kernel-trace --stats` the same script gives the per-kernel view (profiles/collect.sh does both).
Prints one JSON object."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from partsbaseddetector_amd import capi
from partsbaseddetector_amd.model import make_image, make_person_model

im = make_image(0, 640, 480)
rows = []
for K in (1, 2, 6, 8, 12):
    m = make_person_model(K=K)
    m.thresh = 3e38
    row = {"mixtures": K, "filters": 26 * K, "contraction_NxK": 26 * K * 800}
    for name, mode in (("exact_valu", capi.PBD_CONV_EXACT), ("mfma_f32", capi.PBD_CONV_MFMA), ("split_bf16x6", capi.PBD_CONV_SPLIT),
                       ("split_f16x3_opt_in", capi.PBD_CONV_SPLIT_F16)):
        h = capi.Handle(m, conv_mode=mode)
        h.set_profiling(True)
        ms = []
        for i in range(8):
            h.detect(im)
            if i >= 3:
                ms.append(h.stage_ms()["pdf"])
        w = h.work()
        h.close()
        row[name + "_ms"] = round(float(np.median(ms)), 4)
        row[name + "_tflops"] = round(w["F_pdf"] / (np.median(ms) * 1e-3) / 1e12, 2)
        row[name + "_algorithmic_GBps"] = round(w["B_pdf"] / (np.median(ms) * 1e-3) / 1e9, 1)
    ha = capi.Handle(m)                                           # what PBD_CONV_AUTO resolves to for this bank
    row["auto_picks"] = {capi.PBD_CONV_EXACT: "exact", capi.PBD_CONV_MFMA: "mfma", capi.PBD_CONV_SPLIT: "split"}[ha.conv_mode]
    ha.close()
    rows.append(row)
print(json.dumps({"workload": "person 26 x K, 640x480, 46 levels, 140725 cells, 5x5x32 filters", "rows": rows}))
