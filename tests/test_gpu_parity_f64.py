"""GPU parity tests of the PartsBasedDetector<double> instantiation (ros/Node.hpp:121,
cells/detect.cpp:93): libpbd_hip.so with pbd_options.scalar_type = PBD_SCALAR_F64, through the C ABI,
vs the oracle's T = double restatement (oracle/pbd_oracle_T.inc).  Everything is bit-exact: the
double path has no tolerance-based kernel."""
import numpy as np
import pytest

from partsbaseddetector_amd import capi
from partsbaseddetector_amd.model import make_face_like_model, make_image, make_person_model, make_tree_model
from tests.util import assert_candidates_equal

pytestmark = pytest.mark.gpu
F64 = np.float64


def _bits(a):
    return np.ascontiguousarray(a, F64).view(np.uint64)


@pytest.fixture(scope="module")
def h64(gpu_required):
    m = make_tree_model([-1, 0, 1, 1, 0], 3, seed=5)
    h = capi.Handle(m, dtype=F64)
    yield h
    h.close()


def _thresh64(orc, model, im, q=99.5):
    model.thresh = -1e30
    fr = orc.detect(model, im, capacity=1, keep=True, dtype=F64)[4]
    vals = np.concatenate([fr.root(l)[0].ravel() for l in range(fr.nlevels)])
    fr.free()
    return float(np.float32(np.percentile(vals, q)))


# ---------------------------------------------------------------- HOGFeatures<double>
@pytest.mark.parametrize("w,h,cn", [(64, 48, 3), (161, 123, 1), (47, 35, 3), (640, 480, 3), (22, 21, 3)])
def test_hog_f64_bit_exact(h64, orc, w, h, cn):
    im = make_image(3, w, h, cn)
    got, ref = h64.hog(im), orc.hog(im, 4, dtype=F64)
    assert got.dtype == F64 and got.shape == ref.shape
    np.testing.assert_array_equal(_bits(got), _bits(ref))


def test_hog_f64_sbin8_and_differs_from_float(gpu_required, orc):
    m = make_tree_model([-1, 0], 2, seed=3, sbin=8)
    h = capi.Handle(m, dtype=F64)
    im = make_image(4, 320, 240)
    got = h.hog(im)
    np.testing.assert_array_equal(_bits(got), _bits(orc.hog(im, 8, dtype=F64)))
    # it is a different instantiation, not a widened copy of the float result
    assert np.any(got != orc.hog(im, 8).astype(F64))
    h.close()


def test_pyramid_f64_levels_bit_exact(h64, orc):
    im = make_image(5, 200, 150)
    h64.pyramid(im)
    g = h64._geo
    fr = orc.detect(h64.model, im, capacity=1, keep=True, dtype=F64)[4]
    assert g["nlevels"] == fr.nlevels
    for l in range(g["nlevels"]):
        np.testing.assert_array_equal(h64.level_image(l), fr.image(l, 3))
        np.testing.assert_array_equal(_bits(h64.level_features(l)), _bits(fr.feat(l)))
    fr.free()


# ---------------------------------------------------------------- SpatialConvolutionEngine(CV_64F)
@pytest.mark.parametrize("nparts,K,seed", [(3, 3, 11), (5, 4, 12)])
def test_pdf_f64_bit_exact(gpu_required, orc, nparts, K, seed):
    m = make_tree_model([-1] + [0] * (nparts - 1), K, seed=seed)
    h = capi.Handle(m, conv_mode=capi.PBD_CONV_EXACT, dtype=F64)
    h.pyramid(make_image(seed, 120, 90))
    g = h._geo
    h.pdf()
    for l in (0, 3, g["nlevels"] - 1):
        ref = orc.pdf_level(h.level_features(l), m.filtersw, dtype=F64)
        for n in range(len(m.filtersw)):
            np.testing.assert_array_equal(_bits(h.level_response(l, n)), _bits(ref[n]))
    h.close()


def test_pdf_f64_generic_kernel_size(gpu_required, orc):
    m = make_tree_model([-1, 0], 2, seed=31, kh=3, kw=3)   # run-time kernel size: generic filter-bank kernel
    h = capi.Handle(m, dtype=F64)
    h.pyramid(make_image(1, 100, 80))
    h.pdf()
    ref = orc.pdf_level(h.level_features(1), m.filtersw, dtype=F64)
    for n in range(len(m.filtersw)):
        np.testing.assert_array_equal(_bits(h.level_response(1, n)), _bits(ref[n]))
    h.close()


# ---------------------------------------------------------------- DistanceTransform<double>
@pytest.mark.parametrize("rows,cols", [(1, 1), (1, 7), (9, 1), (7, 9), (64, 64), (65, 130), (118, 158), (3, 300)])
def test_dt2d_f64_bit_exact(h64, orc, rows, cols):
    rng = np.random.default_rng(rows * 1000 + cols)
    a = rng.normal(0, 1.5, (rows, cols))
    got = h64.dt2d(a, -0.02, 0.003, -0.03, 0.001, 1, -2)
    ref = orc.dt2d(a, -0.02, 0.003, -0.03, 0.001, 1, -2, dtype=F64)
    np.testing.assert_array_equal(_bits(got[0]), _bits(ref[0]))
    np.testing.assert_array_equal(got[1], ref[1])
    np.testing.assert_array_equal(got[2], ref[2])


def test_dt2d_f64_ties_and_plateaus(h64, orc):
    rng = np.random.default_rng(3)
    for a in (np.zeros((12, 17)), np.round(rng.normal(0, 2, (33, 41))), np.full((5, 64), -3.5),
              np.repeat(rng.normal(0, 1, (20, 1)), 30, axis=1)):
        for (ax, bx, ay, by) in ((-1.0, 0.0, -1.0, 0.0), (-0.01, 0.0, -0.01, 0.0), (-0.5, 0.25, -0.125, -0.5)):
            got = h64.dt2d(a, ax, bx, ay, by, 0, 0)
            ref = orc.dt2d(a, ax, bx, ay, by, 0, 0, dtype=F64)
            np.testing.assert_array_equal(_bits(got[0]), _bits(ref[0]))
            np.testing.assert_array_equal(got[1], ref[1])
            np.testing.assert_array_equal(got[2], ref[2])


# ---------------------------------------------------------------- DynamicProgram<double>::min
def _dp_case64(orc, model, w, h, seed):
    hd = capi.Handle(model, conv_mode=capi.PBD_CONV_EXACT, dtype=F64)
    hd.begin_frame(w, h, 3)
    g = hd._geo
    rng = np.random.default_rng(seed)
    nf = len(model.filtersw)
    desc = model.to_desc()
    resp = [rng.normal(0, 1, (nf, g["cell_h"][l], g["cell_w"][l])) for l in range(g["nlevels"])]
    for l in range(g["nlevels"]):
        for n in range(nf):
            hd.set_level_response(l, n, resp[l][n])
    hd.dp_min()
    for l in range(g["nlevels"]):
        for c in range(model.ncomponents):
            Ix, Iy, Ik, rv, ri = orc.dp_min_level(desc, c, resp[l], dtype=F64)
            grv, gri = hd.root(l, c)
            np.testing.assert_array_equal(_bits(grv), _bits(rv))
            np.testing.assert_array_equal(gri, ri)
            plane = 0
            for p in range(1, model.nparts(c)):
                L = len(model.filterid[c][model.parentid[c][p]])
                for pm in range(L):
                    gx, gy, gk = hd.dp_pointers(l, c, p, pm)
                    np.testing.assert_array_equal(gx, Ix[plane]); np.testing.assert_array_equal(gy, Iy[plane])
                    np.testing.assert_array_equal(gk, Ik[plane])
                    plane += 1
    hd.close()


def test_dp_min_f64_bit_exact_tree(gpu_required, orc):
    _dp_case64(orc, make_tree_model([-1, 0, 1, 1, 0, 4, 4, 2], 3, seed=9), 120, 90, 1)


def test_dp_min_f64_single_mixture_and_multi_component(gpu_required, orc):
    _dp_case64(orc, make_tree_model([-1, 0, 0, 1], 1, seed=10), 90, 70, 2)
    _dp_case64(orc, make_face_like_model(seed=5, ncomp=3, nfilters=20, part_counts=(5, 9)), 80, 60, 3)


# ---------------------------------------------------------------- detect() end to end
def _e2e64(orc, model, im, q=99.5):
    model.thresh = _thresh64(orc, model, im, q)
    ref = orc.detect(model, im, dtype=F64)[:3]
    hd = capi.Handle(model, conv_mode=capi.PBD_CONV_EXACT, dtype=F64)
    got = hd.detect(im)
    hd.close()
    return got, ref


def test_detect_f64_exact_small_tree(gpu_required, orc):
    got, ref = _e2e64(orc, make_tree_model([-1, 0, 1, 1, 0], 3, seed=5), make_image(0, 200, 150))
    assert len(ref[0]) > 5
    assert_candidates_equal(got, ref)


def test_detect_f64_exact_gray_and_face_like(gpu_required, orc):
    got, ref = _e2e64(orc, make_tree_model([-1, 0, 0], 2, seed=6), make_image(1, 161, 131, cn=1))
    assert_candidates_equal(got, ref)
    m = make_face_like_model(seed=8, ncomp=4, nfilters=30, part_counts=(7, 12))
    got, ref = _e2e64(orc, m, make_image(2, 160, 120))
    assert len(ref[0]) > 5
    assert_candidates_equal(got, ref)


def test_detect_f64_person_640x480_exact(gpu_required, orc):
    m = make_person_model(K=2)
    im = make_image(7, 640, 480)
    got, ref = _e2e64(orc, m, im, q=99.9)
    assert len(ref[0]) > 20
    assert_candidates_equal(got, ref)


def test_detect_f64_differs_from_float_instantiation(gpu_required, orc):
    """The two instantiations are different programs: same frame, different low-order score bits."""
    m = make_tree_model([-1, 0, 1, 1, 0], 3, seed=5)
    im = make_image(0, 200, 150)
    m.thresh = -1e30
    h32, h64_ = capi.Handle(m, conv_mode=capi.PBD_CONV_EXACT), capi.Handle(m, dtype=F64)
    h32.pyramid(im); h32.pdf(); h32.dp_min()
    h64_.pyramid(im); h64_.pdf(); h64_.dp_min()
    r32, r64 = h32.root(0, 0)[0], h64_.root(0, 0)[0]
    assert r64.dtype == F64 and r32.dtype == np.float32
    assert np.abs(r64 - r32).max() < 1e-4          # same computation ...
    assert np.any(r64.astype(np.float32) != r32)   # ... carried out in another precision
    h32.close(); h64_.close()


# ---------------------------------------------------------------- fp64 MFMA filter bank (k_conv_mfma_f64)
def test_pdf_f64_mfma_tolerance(gpu_required, orc):
    """v_mfma_f64_16x16x4_f64 accumulates in (half, tap, channel) order: not the reference's order, so it is
    held to a tolerance — 1e-12, eight orders inside the north_star's 1e-4."""
    for nparts, K, seed in ((5, 4, 13), (9, 5, 14)):      # 20 and 45 filters: partial 16-filter n-tiles
        m = make_tree_model([-1] + [0] * (nparts - 1), K, seed=seed)
        h = capi.Handle(m, conv_mode=capi.PBD_CONV_MFMA, dtype=F64)
        h.pyramid(make_image(seed, 120, 90))
        g = h._geo
        h.pdf()
        for l in (0, 3, g["nlevels"] - 1):
            ref = orc.pdf_level(h.level_features(l), m.filtersw, dtype=F64)
            for n in range(len(m.filtersw)):
                assert np.abs(h.level_response(l, n) - ref[n]).max() < 1e-12
        h.close()


def test_detect_f64_mfma_person_matches_oracle(gpu_required, orc):
    """AUTO picks the fp64 MFMA filter bank for the person model; candidates, part locations and boxes
    equal the oracle's, scores to 1e-6 (they are narrowed to float in the candidate record)."""
    m = make_person_model(K=2)
    im = make_image(7, 640, 480)
    m.thresh = _thresh64(orc, m, im, 99.9)
    ref = orc.detect(m, im, dtype=F64)[:3]
    hd = capi.Handle(m, dtype=F64)                         # PBD_CONV_AUTO
    got = hd.detect(im)
    hd.close()
    key = lambda r: sorted(zip(r[0]["level"].tolist(), r[0]["component"].tolist(), map(tuple, r[2].reshape(len(r[0]), -1).tolist())))
    gk, rk = key(got), key(ref)
    common = set(gk) & set(rk)
    assert len(common) >= len(rk) - 2 and len(gk) <= len(rk) + 2      # a score within 1e-13 of the threshold may flip
    if len(gk) == len(rk):
        assert_candidates_equal(got, ref, score_tol=1e-6)


def test_f64_handle_type_checks(gpu_required):
    m = make_tree_model([-1, 0], 2, seed=1)
    h = capi.Handle(m, dtype=F64)
    h.pyramid(make_image(0, 64, 48))
    import ctypes as C
    buf = np.zeros(4096 * 32, np.float32)
    rc = h.L.pbd_get_level_features(h.h, 0, buf.ctypes.data_as(C.POINTER(C.c_float)))   # float getter on a double handle
    assert rc == capi.PBD_ERR_STATE
    assert b"double" in h.L.pbd_last_error(h.h)
    h.close()
    hf = capi.Handle(m)
    hf.pyramid(make_image(0, 64, 48))
    bufd = np.zeros(4096 * 32, np.float64)
    assert hf.L.pbd_get_level_features_f64(hf.h, 0, bufd.ctypes.data_as(C.POINTER(C.c_double))) == capi.PBD_ERR_STATE
    hf.close()


def test_cpp_host_demo_double_matches_oracle(gpu_required, orc, tmp_path):
    """pbd::PartsBasedDetector<double> (host/pbd_host.hpp) driven by the demo call sequence, fused and stage by
    stage — the instantiation ros/Node.hpp:121 and cells/detect.cpp:93 use."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(capi.LIB_PATH), "host", "pbd_demo")
    assert os.path.exists(exe), "build() did not produce the C++ demo"
    m = make_tree_model([-1, 0, 1, 1, 0], 3, seed=5)
    im = make_image(0, 200, 150)
    m.thresh = _thresh64(orc, m, im, 99.5)
    m.save(str(tmp_path / "model.bin"))
    im.tofile(str(tmp_path / "im.raw"))
    heads, boxes, _ = orc.candidates_sort(*orc.detect(m, im, dtype=F64)[:3])
    assert len(heads) > 5
    for extra in ("double", "stagewise-double"):
        out = subprocess.run([exe, str(tmp_path / "model.bin"), str(tmp_path / "im.raw"), "200", "150", "3", extra],
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        lines = out.stdout.strip().splitlines()
        if extra.startswith("stagewise"):   # min()/argmin() with the reference's signatures: their Ix / Iy / Ik tables
            from tests.test_gpu_parity import _tables_checksum
            assert int(lines[0].split()[1]) == _tables_checksum(orc, m, im, dtype=F64)
            lines = lines[1:]
        assert lines[0] == f"Number of candidates: {len(heads)}"
        for ln, h, b in zip(lines[1:], heads, boxes):
            tok = ln.split()
            assert np.float32(float(tok[0])) == h["score"] and int(tok[2]) == h["level"]
            got = np.array([[int(v) for v in t.split(",")] for t in tok[3:]])
            np.testing.assert_array_equal(got, b[: len(got)])


def test_dt2d_f64_random_sweep_all_lane_sharing_modes(h64, orc):
    rng = np.random.default_rng(2027)
    shapes = [(3, 1200), (1200, 2), (9, 300), (40, 200), (200, 40), (33, 120), (64, 90), (17, 60), (25, 45), (12, 30), (7, 10)]
    for i, (r, c) in enumerate(shapes * 2):
        a = rng.normal(0, 1.5, (r, c)) if i % 2 == 0 else np.round(rng.normal(0, 2, (r, c)))
        ax, ay = -float(rng.choice([1.0, 0.5, 0.05, 0.01, 0.003])), -float(rng.choice([1.0, 0.25, 0.02, 0.007]))
        bx, by = float(rng.uniform(-0.05, 0.05)), float(rng.choice([0.0, 0.01, -0.02]))
        osx, osy = int(rng.integers(-4, 5)), int(rng.integers(-4, 5))
        got = h64.dt2d(a, ax, bx, ay, by, osx, osy)
        ref = orc.dt2d(a, ax, bx, ay, by, osx, osy, dtype=F64)
        np.testing.assert_array_equal(_bits(got[0]), _bits(ref[0]), err_msg=f"case {i} {r}x{c}")
        np.testing.assert_array_equal(got[1], ref[1]); np.testing.assert_array_equal(got[2], ref[2])


def test_detect_f64_mfma_person_K6_matches_oracle(gpu_required, orc):
    """Full 26 x 6 person model at 640x480 on the double instantiation with the fp64 MFMA filter bank (what
    PBD_CONV_AUTO selects): candidates, part locations and boxes against orc.detect<double>; differences classified
    like the float case (tests/test_gpu_parity.py::_classified_compare) — with |delta resp| ~1e-13 flips need an
    (essentially exact) tie."""
    from tests.test_gpu_parity import _classified_compare
    m = make_person_model()
    im = make_image(0, 640, 480)
    m.thresh = _thresh64(orc, m, im, 99.9)
    rh, rb, rl, _, fr = orc.detect(m, im, keep=True, dtype=F64)
    hd = capi.Handle(m, dtype=F64)
    got = hd.detect(im)
    n, flips, ties, bugs, worst = _classified_compare(orc, m, im, hd, got, (rh, rb, rl), fr, dtype=F64, tol=1e-6)
    hd.close(); fr.free()
    print(f"person 26x6 640x480 double: {len(rh)} reference candidates, {n} common, {flips} flips, {len(bugs)} bugs")
    assert len(rh) > 50 and n >= len(rh) - 2 and not bugs, bugs
