"""Oracle vs outputs of the REAL reference (oracle/ref_recipe/README.md).  The pins file can only be produced on a
machine with OpenCV 2.4 + Boost + the reference checkout; it does not exist in this repository yet, so every test here
is skipped and parity stays "unpinned" (DESIGN.md §3).  When tests/golden/ref_pins_v1.npz exists, these are the tests
that pin the oracle."""
import os

import numpy as np
import pytest

PINS = os.path.join(os.path.dirname(__file__), "golden", "ref_pins_v1.npz")
pytestmark = pytest.mark.skipif(not os.path.exists(PINS), reason="no reference pins: run oracle/ref_recipe on a machine with OpenCV 2.4")


@pytest.fixture(scope="module")
def pins():
    return np.load(PINS)


def _inputs():
    from partsbaseddetector_amd.model import make_face_like_model, make_image, make_tree_model
    return {"tree": (make_tree_model([-1, 0, 1, 1, 0], 3, seed=5), make_image(0, 200, 150, 3)),
            "gray": (make_tree_model([-1, 0, 0], 2, seed=6), make_image(1, 161, 131, 1)),
            "face": (make_face_like_model(seed=8, ncomp=4, nfilters=30, part_counts=(7, 12)), make_image(2, 160, 120, 3))}


@pytest.mark.parametrize("tag", ["tree", "gray", "face"])
@pytest.mark.parametrize("dtype,sfx", [(np.float32, "f32"), (np.float64, "f64")])
def test_frame_stages_match_the_reference(orc, pins, tag, dtype, sfx):
    model, im = _inputs()[tag]
    model.thresh = 0.0
    fr = orc.detect(model, im, capacity=1 << 20, keep=True, dtype=dtype)
    heads, boxes, locs, _, frame = fr
    key = f"{tag}_{sfx}"
    np.testing.assert_array_equal(pins[f"{key}_scales"].view(np.uint32), orc.geometry(im.shape[1], im.shape[0], model.sbin, model.interval)["scales"].view(np.uint32))
    desc = model.to_desc()
    for l in range(frame.nlevels):
        # image pyramid (cv::resize / cv::pyrDown of the OpenCV that was linked) + HOG: bit for bit
        np.testing.assert_array_equal(pins[f"{key}_feat_{l}"].ravel(), frame.feat(l).ravel())
        ro = frame.resp(l)
        for n in range(ro.shape[0]):
            assert np.abs(pins[f"{key}_resp_{l}_{n}"] - ro[n]).max() < 1e-5
        # min() on the REFERENCE's responses: tables bit for bit
        rr = np.stack([pins[f"{key}_resp_{l}_{n}"] for n in range(ro.shape[0])]).astype(dtype)
        for c in range(model.ncomponents):
            Ix, Iy, Ik, rv, ri = orc.dp_min_level(desc, c, rr, dtype=dtype)
            np.testing.assert_array_equal(pins[f"{key}_rootv_{l}_{c}"], rv)
            np.testing.assert_array_equal(pins[f"{key}_rooti_{l}_{c}"], ri)
            plane = 0
            for p in range(1, model.nparts(c)):
                for m in range(len(model.filterid[c][model.parentid[c][p]])):
                    np.testing.assert_array_equal(pins[f"{key}_Ix_{l}_{c}_{p}_{m}"], Ix[plane])
                    np.testing.assert_array_equal(pins[f"{key}_Iy_{l}_{c}_{p}_{m}"], Iy[plane])
                    np.testing.assert_array_equal(pins[f"{key}_Ik_{l}_{c}_{p}_{m}"], Ik[plane])
                    plane += 1
    frame.free()


@pytest.mark.parametrize("tag,kind,shape,seed", [("w16u", np.uint16, (150, 110, 3), 10), ("w32f", np.float32, (97, 131, 1), 11), ("w64f", np.float64, (203, 77, 3), 12)])
@pytest.mark.parametrize("dtype,sfx", [(np.float32, "f32"), (np.float64, "f64")])
def test_wide_depth_feature_pyramids_match_the_reference(orc, pins, tag, kind, shape, seed, dtype, sfx):
    """CV_16U / CV_32F / CV_64F images (src/HOGFeatures.cpp:136-146): pins cv::resize / cv::pyrDown of those depths (the restatement follows
    OpenCV 2.4's scalar loops: a build whose pyrDown takes the SSE row pass may differ in the last bit for float images — then THIS is where it
    shows) and features<IT>"""
    from partsbaseddetector_amd.model import make_tree_model, make_wide_image
    if f"{tag}_{sfx}_feat_0" not in pins:
        pytest.skip("pins file predates the wide-depth frames")
    w, h, cn = shape
    im = make_wide_image(kind, seed, w, h, cn)
    m = make_tree_model([-1, 0], 1, seed=1)                      # sbin 4, interval 10 (what dump_reference.cpp's run_wide passes): only the pyramid is compared
    assert (m.sbin, m.interval) == (4, 10)
    m.thresh = 3e38
    _, _, _, _, frame = orc.detect(m, im, capacity=1, keep=True, dtype=dtype)
    for l in range(frame.nlevels):
        np.testing.assert_array_equal(pins[f"{tag}_{sfx}_feat_{l}"].ravel(), frame.feat(l).ravel())
    frame.free()


@pytest.mark.parametrize("dtype,sfx", [(np.float32, "f32"), (np.float64, "f64")])
def test_distance_transform_matches_the_reference(orc, pins, dtype, sfx):
    rng = np.random.default_rng(20260927)
    for i, (r, c) in enumerate([(7, 9), (23, 31), (40, 57), (118, 158), (1, 7), (9, 1)]):
        a = rng.normal(0, 1.5, (r, c)).astype(np.float32)
        if i == 2:
            a = np.round(a)
        out, ix, iy = orc.dt2d(a.astype(dtype), -0.01 - 0.01 * i, 0.002 * i, -0.02, -0.001 * i, i % 5 - 2, 2 - i % 5, dtype=dtype)
        np.testing.assert_array_equal(pins[f"dt_{sfx}_out_{i}"], out)
        np.testing.assert_array_equal(pins[f"dt_{sfx}_ix_{i}"], ix)
        np.testing.assert_array_equal(pins[f"dt_{sfx}_iy_{i}"], iy)
