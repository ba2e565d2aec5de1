"""CPU checks of the arithmetic the split-product filter bank (partsbaseddetector_amd/csrc/k_conv_split.hip) rests on — independent of any GPU:
an fp32 number is exactly the sum of three bfloat16 (round-to-nearest-even splits, the bit manipulation of `bf16_rn_bits`), every product of two
parts is exact in fp32, and the six partial products above 2^-24 relative, accumulated in fp32, reproduce a K = 800 dot product (one response of the
5 x 5 x 32 bank) with an error of the size of a plain fp32 chain's."""
import numpy as np


def bf16_rn_bits(u):
    """fp32 bit patterns (uint32) -> bfloat16 bit patterns, round to nearest even: the kernel's `(u + 0x7FFF + ((u >> 16) & 1)) >> 16`"""
    u = u.astype(np.uint64)
    return ((u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) >> np.uint64(16)).astype(np.uint32)


def split3(x):
    """x (float32) -> three float32 arrays that are exact bfloat16 values with x == h + m + l"""
    parts, r = [], x.astype(np.float32).copy()
    for _ in range(3):
        p = (bf16_rn_bits(r.view(np.uint32)) << np.uint32(16)).view(np.float32)
        parts.append(p)
        r = (r - p).astype(np.float32)              # exact: the difference has at most 16 (then 8) significant bits
    return parts, r


def _samples(rng, n):
    mag = np.exp2(rng.uniform(-60, 40, n))
    x = (mag * rng.choice([-1.0, 1.0], n)).astype(np.float32)
    u = x.view(np.uint32).copy()
    pick = rng.random(n) < 0.3                       # mantissas next to the bfloat16 rounding boundaries
    u[pick] = (u[pick] & np.uint32(0xFFFF0000)) | rng.choice(np.array([0x7FFF, 0x8000, 0x8001, 0x7F80, 0x807F, 0xFFFF, 0x0001], np.uint32), int(pick.sum()))
    x = u.view(np.float32)
    x[:8] = [0.0, -0.0, 1.0, -1.0, 0.2, 0.5, 3.0e38, -3.0e38]
    return x[np.isfinite(x)]


def test_three_bfloat16_parts_are_exact():
    rng = np.random.default_rng(0)
    x = _samples(rng, 200000)
    (h, m, l), rest = split3(x)
    assert np.all(rest == 0)                                                        # nothing left after three parts
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64), x.astype(np.float64))
    for p in (h, m, l):
        assert np.all((p.view(np.uint32) & np.uint32(0xFFFF)) == 0)                 # each part is a bfloat16
    nz = h != 0
    assert np.all(np.abs(m[nz]) <= np.abs(h[nz]) * 2.0 ** -8) and np.all(np.abs(l[nz]) <= np.abs(h[nz]) * 2.0 ** -16)


def test_products_of_parts_are_exact_in_fp32():
    rng = np.random.default_rng(1)
    a, b = _samples(rng, 100000), _samples(rng, 100000)
    n = min(len(a), len(b))
    a, b = a[:n] * np.float32(2.0 ** -20), b[:n] * np.float32(2.0 ** -20)           # (keep the products inside the fp32 range)
    pa, _ = split3(a)
    pb, _ = split3(b)
    for x in pa:
        for y in pb:
            exact = x.astype(np.float64) * y.astype(np.float64)
            ok = ((np.abs(exact) >= 2.0 ** -126) & (np.abs(exact) < 2.0 ** 127)) | (exact == 0)   # (subnormal / overflowing products: outside the claim)
            with np.errstate(over="ignore"):
                got = (x * y).astype(np.float64)
            assert np.array_equal(got[ok], exact[ok])                               # 8 x 8 significant bits fit fp32's 24


def test_six_partial_products_match_an_fp32_chain():
    rng = np.random.default_rng(2)
    K, N = 800, 400
    f = np.abs(rng.normal(0.08, 0.06, (N, K))).astype(np.float32)                   # HOG-like features
    w = rng.normal(0.0, 0.02, (N, K)).astype(np.float32)
    ref = (f.astype(np.float64) * w.astype(np.float64)).sum(1)
    chain = np.zeros(N, np.float32)
    for k in range(K):                                                              # plain fp32 chain (product rounded, sum rounded: no better than an fma chain)
        chain = (chain + f[:, k] * w[:, k]).astype(np.float32)
    pf, _ = split3(f)
    pw, _ = split3(w)
    acc = np.zeros(N, np.float32)
    for k0 in range(0, K, 16):                                                      # one MFMA k-step = 16 channels; products in the kernel's order
        for sa, sb in ((1, 1), (0, 2), (2, 0), (0, 1), (1, 0), (0, 0)):
            part = (pf[sa][:, k0:k0 + 16].astype(np.float64) * pw[sb][:, k0:k0 + 16].astype(np.float64)).sum(1)   # exact products, summed inside the instruction
            acc = (acc.astype(np.float64) + part).astype(np.float32)                # one fp32 rounding per MFMA
    e_split, e_chain = np.abs(acc - ref).max(), np.abs(chain - ref).max()
    dropped = np.abs((pf[1].astype(np.float64) * pw[2] + pf[2].astype(np.float64) * pw[1] + pf[2].astype(np.float64) * pw[2]).sum(1)).max()
    assert dropped < 2.0 ** -22 * np.abs(ref).max() + 1e-9                          # the three products left out are below fp32's resolution of the result
    assert e_split <= 1.5 * e_chain + 1e-7, (e_split, e_chain)
    assert e_split < 2e-6


# ---- PBD_CONV_SPLIT_F16: two scaled binary16 parts, three products ----
def split2_f16(x, e):
    """x (float32), e (exponent) -> two float32 arrays holding binary16 values with x 2^e ~= h + m (k_feat_split16 / conv_split16_filters)"""
    s = np.ldexp(x.astype(np.float32), e).astype(np.float32)
    h = s.astype(np.float16).astype(np.float32)                                     # round to nearest even, subnormals kept
    m = (s - h).astype(np.float32).astype(np.float16).astype(np.float32)            # (the subtraction is exact)
    return h, m, (s - h - m).astype(np.float32)


def test_two_binary16_parts_hold_23_bits():
    rng = np.random.default_rng(3)
    x = np.exp2(rng.uniform(-12, 0, 200000)).astype(np.float32) * rng.choice([-1.0, 1.0], 200000).astype(np.float32)
    u = x.view(np.uint32).copy()
    pick = rng.random(len(x)) < 0.3                  # mantissas next to the binary16 rounding boundaries
    u[pick] = (u[pick] & np.uint32(0xFFFFE000)) | rng.choice(np.array([0x0FFF, 0x1000, 0x1001, 0x1FFF], np.uint32), int(pick.sum()))
    x = u.view(np.float32)
    h, m, rest = split2_f16(x, 12)                   # features: scaled by 2^12
    s = np.ldexp(x, 12)
    assert np.all(np.abs(rest) <= np.abs(s) * 2.0 ** -23 + 2.0 ** -25)              # 23 bits, or the subnormal spacing's half below 2^-14
    assert np.all(np.abs(m) <= np.abs(h) * 2.0 ** -11 + 2.0 ** -25)
    b, bm, _ = split2_f16(x[::-1].copy(), 14)
    for p, q in ((h, b), (h, bm), (m, b)):                                          # 11 x 11 significant bits: exact in fp32
        assert np.array_equal((p * q).astype(np.float64), p.astype(np.float64) * q.astype(np.float64))
    tiny = np.float32(2.0 ** -30) * x                                               # far below the scaled range: absolute, not relative, precision
    _, _, rest = split2_f16(tiny, 12)
    assert np.all(np.abs(rest) <= 2.0 ** -25)


def test_three_binary16_products_match_an_fp32_chain():
    rng = np.random.default_rng(4)
    K, N = 800, 400
    f = np.abs(rng.normal(0.08, 0.06, (N, K))).astype(np.float32)
    w = rng.normal(0.0, 0.02, (N, K)).astype(np.float32)
    ref = (f.astype(np.float64) * w.astype(np.float64)).sum(1)
    chain = np.zeros(N, np.float32)
    for k in range(K):
        chain = (chain + f[:, k] * w[:, k]).astype(np.float32)
    we = 14 - (np.frexp(np.abs(w).max(1))[1])                                       # per filter (row): max |w| 2^e in [2^13, 2^14)
    fh, fm, _ = split2_f16(f, 12)
    wh, wm, _ = split2_f16(w, we[:, None])
    acc = np.zeros(N, np.float32)
    for k0 in range(0, K, 16):
        for a, b in ((fh, wm), (fm, wh), (fh, wh)):                                 # the kernel's order: the small products of a k-step first
            part = (a[:, k0:k0 + 16].astype(np.float64) * b[:, k0:k0 + 16].astype(np.float64)).sum(1)
            acc = (acc.astype(np.float64) + part).astype(np.float32)
    got = np.ldexp(acc, -(12 + we)).astype(np.float32)                              # exact
    e_split, e_chain = np.abs(got - ref).max(), np.abs(chain - ref).max()
    assert e_split <= 1.5 * e_chain + 1e-7, (e_split, e_chain)
    assert e_split < 2e-6
