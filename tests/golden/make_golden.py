"""Generate the committed regression fixtures in tests/golden/ from the CPU oracle.

The reference ships no golden vectors for this path (test/CMakeLists.txt:1-10) and cannot be built
in this image (OpenCV/Boost absent), so these are REGRESSION vectors of oracle/pbd_oracle.c — they
pin the oracle against accidental change and give the GPU tests a fixed target; they do not pin
the oracle to the reference ("parity unpinned").

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402
from partsbaseddetector_amd.model import make_face_like_model, make_image, make_tree_model  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.default_rng(20260927)
    g = {}
    # distance transform: a few shapes, incl. ties
    for i, (r, c) in enumerate([(7, 9), (23, 31), (40, 57)]):
        a = rng.normal(0, 1.5, (r, c)).astype(np.float32)
        if i == 2:
            a = np.round(a)
        par = (-0.01 - 0.01 * i, 0.002 * i, -0.02, -0.001 * i, i - 1, 1 - i)
        out, ix, iy = orc.dt2d(a, *par)
        g[f"dt{i}_in"], g[f"dt{i}_par"] = a, np.asarray(par, np.float64)
        g[f"dt{i}_out"], g[f"dt{i}_ix"], g[f"dt{i}_iy"] = out, ix.astype(np.int16), iy.astype(np.int16)
    # image pyramid + HOG
    im = make_image(11, 72, 56)
    g["im"] = im
    g["resize_45x35"] = orc.resize(im, 45, 35)
    g["pyrdown"] = orc.pyrdown(im)
    g["hog_sbin4"] = orc.hog(im, 4)
    g["hog_gray_sbin4"] = orc.hog(np.ascontiguousarray(im[..., 1]), 4)
    # pdf
    m = make_tree_model([-1, 0, 0], 2, seed=42)
    feat = orc.hog(make_image(12, 60, 48), 4)
    g["pdf_feat"] = feat
    g["pdf_resp"] = orc.pdf_level(feat, m.filtersw)
    # dp min on a small tree
    resp = rng.normal(0, 1, (len(m.filtersw), 11, 14)).astype(np.float32)
    Ix, Iy, Ik, rv, ri = orc.dp_min_level(m.to_desc(), 0, resp)
    g["dp_resp"], g["dp_ix"], g["dp_iy"], g["dp_ik"] = resp, Ix.astype(np.int16), Iy.astype(np.int16), Ik.astype(np.int8)
    g["dp_rootv"], g["dp_rooti"] = rv, ri.astype(np.int8)
    # end to end: tree + face-like
    for name, model, img in (("tree", make_tree_model([-1, 0, 1, 1, 0], 3, seed=5), make_image(0, 120, 90)),
                             ("face", make_face_like_model(seed=8, ncomp=3, nfilters=24, part_counts=(6, 9)), make_image(2, 110, 84))):
        model.thresh = -1e30
        _, _, _, _, fr = orc.detect(model, img, capacity=1, keep=True)
        vals = np.concatenate([fr.root(l)[0].ravel() for l in range(fr.nlevels)])
        fr.free()
        model.thresh = float(np.float32(np.percentile(vals, 99.0)))
        heads, boxes, locs, _ = orc.detect(model, img)
        g[f"e2e_{name}_thresh"] = np.float32(model.thresh)
        g[f"e2e_{name}_heads"] = np.stack([heads["score"].view(np.int32), heads["component"], heads["level"], heads["nparts"]], 1)
        g[f"e2e_{name}_boxes"], g[f"e2e_{name}_locs"] = boxes.astype(np.int16), locs.astype(np.int16)
    np.savez_compressed(os.path.join(OUT, "golden_v1.npz"), **g)
    print("wrote", os.path.join(OUT, "golden_v1.npz"), sum(v.nbytes for v in g.values()), "bytes raw")


def main_f64():
    """Regression vectors of the T = double instantiation (oracle/pbd_oracle_T.inc with T = double)."""
    F = np.float64
    rng = np.random.default_rng(20260928)
    g = {}
    for i, (r, c) in enumerate([(7, 9), (23, 31), (40, 57)]):
        a = rng.normal(0, 1.5, (r, c))
        if i == 2:
            a = np.round(a)
        par = (-0.01 - 0.01 * i, 0.002 * i, -0.02, -0.001 * i, i - 1, 1 - i)
        out, ix, iy = orc.dt2d(a, *par, dtype=F)
        g[f"dt{i}_in"], g[f"dt{i}_par"] = a, np.asarray(par, F)
        g[f"dt{i}_out"], g[f"dt{i}_ix"], g[f"dt{i}_iy"] = out, ix.astype(np.int16), iy.astype(np.int16)
    im = make_image(11, 72, 56)
    g["im"] = im
    g["hog_sbin4"] = orc.hog(im, 4, dtype=F)
    g["hog_gray_sbin4"] = orc.hog(np.ascontiguousarray(im[..., 1]), 4, dtype=F)
    m = make_tree_model([-1, 0, 0], 2, seed=42)
    feat = orc.hog(make_image(12, 60, 48), 4, dtype=F)
    g["pdf_resp"] = orc.pdf_level(feat, m.filtersw, dtype=F)
    resp = rng.normal(0, 1, (len(m.filtersw), 11, 14))
    Ix, Iy, Ik, rv, ri = orc.dp_min_level(m.to_desc(), 0, resp, dtype=F)
    g["dp_resp"], g["dp_ix"], g["dp_iy"], g["dp_ik"] = resp, Ix.astype(np.int16), Iy.astype(np.int16), Ik.astype(np.int8)
    g["dp_rootv"], g["dp_rooti"] = rv, ri.astype(np.int8)
    model, img = make_tree_model([-1, 0, 1, 1, 0], 3, seed=5), make_image(0, 120, 90)
    model.thresh = -1e30
    fr = orc.detect(model, img, capacity=1, keep=True, dtype=F)[4]
    vals = np.concatenate([fr.root(l)[0].ravel() for l in range(fr.nlevels)])
    fr.free()
    model.thresh = float(np.float32(np.percentile(vals, 99.0)))
    heads, boxes, locs, _ = orc.detect(model, img, dtype=F)
    g["e2e_tree_thresh"] = np.float32(model.thresh)
    g["e2e_tree_heads"] = np.stack([heads["score"].view(np.int32), heads["component"], heads["level"], heads["nparts"]], 1)
    g["e2e_tree_boxes"], g["e2e_tree_locs"] = boxes.astype(np.int16), locs.astype(np.int16)
    np.savez_compressed(os.path.join(OUT, "golden_f64_v1.npz"), **g)
    print("wrote", os.path.join(OUT, "golden_f64_v1.npz"), sum(v.nbytes for v in g.values()), "bytes raw")


if __name__ == "__main__":
    if "--f64-only" not in sys.argv:
        main()
    main_f64()
