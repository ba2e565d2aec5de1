"""Ad-hoc stress run (not a test): the split-product filter bank (PBD_CONV_SPLIT, k_conv_split32) on random banks — 16 .. 340 filters (every
n-tile remainder, one to three n-tile groups), 3x3 .. 9x9 and rectangular sizes, cell sizes 4 / 8, random image sizes (ragged tiles on every level,
levels smaller than a tile), gray / colour, single frames and batches — against the oracle's tap-ordered fp32 sums on every level, and against an
fp64 correlation on a sample; PBD_CONV_AUTO must have resolved to the split bank.  `f16`: the same run on the opt-in PBD_CONV_SPLIT_F16 bank.

    python tests/tools_fuzz_split.py [seconds] [seed] [f16]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import orc  # noqa: E402
from partsbaseddetector_amd import capi  # noqa: E402
from partsbaseddetector_amd.model import make_image, make_tree_model  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    f16 = len(sys.argv) > 3 and sys.argv[3] == "f16"
    rng = np.random.default_rng(seed)
    t0 = time.time()
    ncase = nplanes = 0
    worst = worst64 = 0.0
    while time.time() - t0 < budget:
        nfilt = int(rng.choice([16, 17, 31, 32, 33, 63, 64, 65, 96, 127, 129, 156, 160, 161, 200, 321, 340]))
        kh, kw = (int(rng.integers(3, 10)),) * 2 if rng.random() < 0.7 else (int(rng.integers(3, 10)), int(rng.integers(3, 10)))
        m = make_tree_model([-1] + [0] * (nfilt - 1), 1, seed=int(rng.integers(1 << 30)), kh=kh, kw=kw, sbin=int(rng.choice([4, 4, 8])),
                            interval=int(rng.choice([3, 5, 10])))
        w, h = int(rng.integers(40, 360)), int(rng.integers(40, 280))
        cn = 1 if rng.random() < 0.2 else 3
        im = make_image(int(rng.integers(1 << 30)), w, h, cn)
        try:
            hd = capi.Handle(m, conv_mode=capi.PBD_CONV_SPLIT_F16 if f16 else capi.PBD_CONV_AUTO)
            hd.pyramid(im)
        except capi.PbdError:                                      # (image too small for the pyramid)
            continue
        assert hd.conv_mode == (capi.PBD_CONV_SPLIT_F16 if f16 else capi.PBD_CONV_SPLIT)
        hd.pdf()
        g = hd._geo
        tol = 2e-5 * max(1.0, kh * kw / 25.0)                      # the contraction is kh kw 32 deep: the bound of the fp32 MFMA bank's tests
        for l in range(g["nlevels"]):
            if g["cell_w"][l] == 0 or g["cell_h"][l] == 0:
                continue
            f = hd.level_features(l)
            ref = orc.pdf_level(f, m.filtersw)
            pick = sorted(set([0, nfilt - 1] + [int(x) for x in rng.integers(0, nfilt, 6)]))
            for n in pick:
                got = hd.level_response(l, n)
                e = float(np.abs(got - ref[n]).max())
                worst = max(worst, e)
                assert e < tol, (ncase, nfilt, kh, kw, w, h, cn, l, n, e)
                nplanes += 1
            if l in (0, g["nlevels"] // 2):
                ref64 = orc.pdf_level(f, [m.filtersw[n] for n in pick[:3]], dtype=np.float64)
                for i, n in enumerate(pick[:3]):
                    worst64 = max(worst64, float(np.abs(hd.level_response(l, n) - ref64[i]).max()))
        hd.close()
        ncase += 1
    print(f"split fuzz ok ({'binary16 x3' if f16 else 'bfloat16 x6'}): {ncase} random banks, {nplanes} response planes within the tolerance of the oracle's fp32 sums (worst {worst:.2e}); "
          f"against fp64 on the sampled planes: worst {worst64:.2e}; {time.time() - t0:.0f} s, seed {seed}")


if __name__ == "__main__":
    main()
