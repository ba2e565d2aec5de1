"""Shared helpers for the parity tests."""
import numpy as np

from partsbaseddetector_amd import capi
from partsbaseddetector_amd.model import make_image, make_tree_model, make_person_model, make_face_like_model


def thresh_from_oracle(orc, model, im, q=99.5):
    """Pick a threshold that yields a few dozen candidates: percentile of the oracle's root scores."""
    model.thresh = -1e30
    _, _, _, _, fr = orc.detect(model, im, capacity=1, keep=True)
    vals = np.concatenate([fr.root(l)[0].ravel() for l in range(fr.nlevels)])
    fr.free()
    return float(np.float32(np.percentile(vals, q)))


def assert_candidates_equal(a, b, score_tol=0.0):
    ha, ba, la = a
    hb, bb, lb = b
    assert len(ha) == len(hb), (len(ha), len(hb))
    for k in ("component", "level", "nparts"):
        np.testing.assert_array_equal(ha[k], hb[k])
    if score_tol == 0.0:
        np.testing.assert_array_equal(ha["score"], hb["score"])
    else:
        np.testing.assert_allclose(ha["score"], hb["score"], atol=score_tol, rtol=0)
    np.testing.assert_array_equal(la, lb)
    np.testing.assert_array_equal(ba, bb)
