"""Ad-hoc study (not a test, CPU only): how accurate would the filter bank be if its fp32 products went through EXACT bf16 splits
(x = h + m + l, three bfloat16 of 8 significant bits each: the 24 bits of an fp32 significand) and fp32 accumulation — i.e. through
the bf16 matrix units (2.5 PFLOP/s on MI355X, 16x the fp32 MFMA rate, and NOT the lanes the vector ALU shares with fp32 MFMAs) —
against fp64, next to plain fp32 accumulation in k-steps of 4 (what the fp32 MFMA path of k_conv_mfma16 does)?  Person bank
(156 5x5x32 filters), real HOG features of 640x480 frames (oracle), interior cells of level 0.  See DESIGN.md section 8.

    python tests/tools_split_products_study.py
"""
import sys, numpy as np
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import orc
from partsbaseddetector_amd.model import make_image, make_person_model

def bf16(x):   # round-to-nearest-even float32 -> bfloat16 (returned as float32)
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)

def split(x, n):
    parts, r = [], np.asarray(x, np.float32).copy()
    for _ in range(n):
        p = bf16(r); parts.append(p); r = (r - p).astype(np.float32)
    return parts

def mm32(A, B, chunk=32):
    """fp32 accumulation in K-chunks (like an MFMA accumulating 32 k at a time: exact products, fp32 running sum)"""
    acc = np.zeros((A.shape[0], B.shape[1]), np.float32)
    for k0 in range(0, A.shape[1], chunk):
        acc = (acc + (A[:, k0:k0+chunk].astype(np.float64) @ B[k0:k0+chunk].astype(np.float64)).astype(np.float32)).astype(np.float32)
    return acc

m = make_person_model(K=6)
W, H = 640, 480
rows = []
for seed in (0, 1, 2):
    im = make_image(seed, W, H)
    feat = orc.hog(im, m.sbin)                        # level 0 of the first octave is the image itself at sbin 4
    Hc, Wc, F = feat.shape
    kh = 5; kw = 5
    filt = np.stack(m.filtersw).reshape(-1, kh, kw, F)          # [nf, kh, kw, 32]
    nf = filt.shape[0]
    # the reference pads with zeros and a 1 in the truncation channel outside; interior cells only here (the study is about products)
    ys, xs = np.arange(2, Hc - 2), np.arange(2, Wc - 2)
    sel = [(y, x) for y in ys[::3] for x in xs[::3]]
    A = np.stack([feat[y-2:y+3, x-2:x+3, :].reshape(-1) for y, x in sel]).astype(np.float32)     # [cells, 800]
    B = filt.reshape(nf, -1).T.astype(np.float32).copy()                                          # [800, nf]
    ref = A.astype(np.float64) @ B.astype(np.float64)
    out = {}
    out['fp32 accumulate (fp32 MFMA-like)'] = mm32(A, B, 4)
    for ns, keep in ((2, [(0,0),(0,1),(1,0)]), (2, [(0,0),(0,1),(1,0),(1,1)]), (3, [(0,0),(0,1),(1,0),(1,1),(0,2),(2,0)]), (3, [(i,j) for i in range(3) for j in range(3)])):
        As, Bs = split(A, ns), split(B, ns)
        acc = np.zeros_like(ref, dtype=np.float32)
        # small terms first, as a kernel would order them to lose least
        for (i, j) in sorted(keep, key=lambda t: -(t[0] + t[1])):
            acc = (acc + mm32(As[i], Bs[j], 32)).astype(np.float32)
        out[f'bf16 {ns}-way split, {len(keep)} products'] = acc
    # PBD_CONV_SPLIT_F16: two binary16 parts of the operands scaled into binary16's range (features 2^12, every filter to max |w| 2^e in [2^13, 2^14)), three products
    f16 = lambda x: np.asarray(x, np.float32).astype(np.float16).astype(np.float32)
    def split16(x):
        h = f16(x); return [h, f16((x - h).astype(np.float32))]
    we = (14 - np.frexp(np.abs(B).max(0))[1]).astype(np.int32)
    As, Bs = split16(np.ldexp(A, 12).astype(np.float32)), split16(np.ldexp(B, we[None, :]).astype(np.float32))
    acc = np.zeros_like(ref, dtype=np.float32)
    for (i, j) in ((0, 1), (1, 0), (0, 0)):
        acc = (acc + mm32(As[i], Bs[j], 32)).astype(np.float32)
    out['binary16 2-way split (scaled), 3 products'] = np.ldexp(acc, -(12 + we)[None, :]).astype(np.float32)
    print(f'seed {seed}: {A.shape[0]} cells x {nf} filters, |response| max {np.abs(ref).max():.3f} rms {np.sqrt((ref**2).mean()):.3f}')
    for k, v in out.items():
        d = np.abs(v.astype(np.float64) - ref)
        print(f'   {k:46s} max |err| {d.max():.3e}   rms {np.sqrt((d**2).mean()):.3e}')
