"""Ad-hoc stress run (not a test): random tree models / image sizes / thresholds through the product path (exact filter bank) against the
oracle, bit for bit — candidates, part locations, boxes and scores — plus random stand-alone 2-D distance transforms (pbd_dt2d) with
adversarial maps (ties, plateaus, weak curvature).  Exercises the round-4 rewrites (dt_core.hpp scan / stitch, fold loader, k_root
block table, HOG / pyramid kernels) far beyond the fixed cases of tests/test_gpu_parity.py.

    python tests/tools_fuzz_detect.py [seconds] [seed]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import orc  # noqa: E402
from partsbaseddetector_amd import capi  # noqa: E402
from partsbaseddetector_amd.model import make_image, make_tree_model  # noqa: E402


def random_tree(rng, n):
    return [-1] + [int(rng.integers(0, p)) for p in range(1, n)]


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0 = time.time()
    ncase = ncand = ndt = 0
    while time.time() - t0 < budget:
        # ---- a random detector
        nparts = int(rng.integers(1, 12))
        K = int(rng.integers(1, 7))
        ksz = int(rng.choice([3, 5, 5, 5, 7]))
        dtype = np.float64 if rng.random() < 0.25 else np.float32
        m = make_tree_model(random_tree(rng, nparts), K, seed=int(rng.integers(1 << 30)), kh=ksz, kw=ksz,
                            sbin=int(rng.choice([4, 4, 8])), interval=int(rng.choice([3, 5, 10])))
        w, h = int(rng.integers(60, 330)), int(rng.integers(60, 250))
        cn = 1 if rng.random() < 0.2 else 3
        im = make_image(int(rng.integers(1 << 30)), w, h, cn)
        try:
            m.thresh = -1e30
            fr = orc.detect(m, im, capacity=1, keep=True, dtype=dtype)[4]
        except Exception as e:            # (geometry the reference rejects, e.g. too few levels)
            continue
        vals = np.concatenate([fr.root(l)[0].ravel() for l in range(fr.nlevels)] or [np.zeros(1)])
        fr.free()
        if vals.size < 10:
            continue
        m.thresh = float(np.float32(np.percentile(vals, float(rng.choice([90.0, 97.0, 99.5])))))
        ref = orc.detect(m, im, capacity=32768, dtype=dtype)[:3]
        dp_mode = int(rng.choice([0, 0, 1, 2]))
        hd = capi.Handle(m, conv_mode=capi.PBD_CONV_EXACT, max_candidates=32768, dtype=dtype, dp_mode=dp_mode, graph=int(rng.integers(0, 2)))
        for rep in range(2):
            got = hd.detect(im, capacity=32768)
            assert len(got[0]) == len(ref[0]), (ncase, nparts, K, w, h, cn, dtype, dp_mode, len(got[0]), len(ref[0]))
            for k in ("component", "level", "nparts"):
                assert np.array_equal(got[0][k], ref[0][k]), (ncase, k)
            assert np.array_equal(got[0]["score"].view(np.uint32), ref[0]["score"].view(np.uint32)), (ncase, "score bits")
            assert np.array_equal(got[2], ref[2]) and np.array_equal(got[1], ref[1]), (ncase, "locations / boxes")
        if rng.random() < 0.3 and dp_mode != 1:            # the same frames as a batch
            outs = hd.detect_batch([im, im, im], capacity=32768)
            for o in outs:
                assert np.array_equal(o[0]["score"].view(np.uint32), ref[0]["score"].view(np.uint32)) and np.array_equal(o[2], ref[2])
        ncand += len(ref[0])
        ncase += 1
        # ---- random stand-alone distance transforms on this handle
        for _ in range(3):
            rows, cols = int(rng.integers(1, 200)), int(rng.integers(1, 300))
            kind = int(rng.integers(0, 5))
            if kind == 0:
                a = rng.normal(0, 1.5, (rows, cols))
            elif kind == 1:
                a = np.round(rng.normal(0, 2, (rows, cols)))                       # exact ties
            elif kind == 2:
                a = np.sin(np.arange(cols) / 7.0)[None, :] + 0.05 * rng.normal(0, 1, (rows, cols))
            elif kind == 3:
                a = (rng.random((rows, cols)) < 0.03) * 5.0
            else:
                a = np.zeros((rows, cols))
            a = a.astype(dtype)
            ax = -float(np.float32(rng.choice([1.0, 0.05, 0.01, 0.003, 0.0001])))
            ay = -float(np.float32(rng.uniform(0.005, 0.05)))
            bx, by = -float(np.float32(rng.uniform(-0.01, 0.01))), 0.0
            osx, osy = int(rng.integers(-4, 5)), int(rng.integers(-4, 5))
            o_ref = orc.dt2d(a, ax, bx, ay, by, osx, osy, dtype=dtype)
            o_got = hd.dt2d(a, ax, bx, ay, by, osx, osy)
            it = np.uint64 if dtype == np.float64 else np.uint32
            assert np.array_equal(np.asarray(o_got[0]).view(it), np.asarray(o_ref[0]).view(it)), (ncase, "dt2d values", rows, cols, kind)
            assert np.array_equal(o_got[1], o_ref[1]) and np.array_equal(o_got[2], o_ref[2]), (ncase, "dt2d pointers", rows, cols, kind)
            ndt += 1
        hd.close()
    print(f"fuzz ok: {ncase} random detectors ({ncand} candidates, every score / location / box bit-identical to the oracle), "
          f"{ndt} random 2-D distance transforms, {time.time() - t0:.0f} s, seed {seed}")


if __name__ == "__main__":
    main()
