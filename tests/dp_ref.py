"""Test helper: the message passing of DynamicProgram<T>::min (src/DynamicProgram.cpp:95-171) for ONE pyramid
level, in numpy on top of the oracle's distance transform (orc.dt2d), keeping every intermediate map — the
accumulated part scores and the distance-transformed child scores that the oracle's C entry point does not
return.  Used to CLASSIFY part-location differences between the MFMA filter bank and the reference-order
filter bank as near-ties: at the first part where two back-tracked configurations diverge, how
far apart were the two alternatives in the reference's own numbers?

Checked against orc.dp_min_level bit for bit (tests/test_oracle_cpu.py::test_dp_ref_matches_oracle).
"""
import numpy as np


def level_maps(orc, model, comp, resp, dtype=np.float32):
    """resp [nf, H, W] -> dict(score_in[p][mm], sdt[p][mm], weighted-argmax inputs, rootv, rooti)."""
    T = np.dtype(dtype).type
    P = model.nparts(comp)
    fid, did, bid, par = model.filterid[comp], model.defid[comp], model.biasid[comp], model.parentid[comp]
    acc = {}                                     # ncscores[filterid] (:93): filled lazily (:155)
    score_in = [None] * P
    sdt = [None] * P
    ix = [None] * P
    iy = [None] * P
    for p in range(P - 1, 0, -1):                # :95
        K = len(fid[p])
        score_in[p], sdt[p], ix[p], iy[p] = [], [], [], []
        for mm in range(K):
            src = acc.get(fid[p][mm], resp[fid[p][mm]])          # :115-119
            w = model.defw[did[p][mm]]
            a = model.anchors[did[p][mm]]
            out, x_, y_ = orc.dt2d(src, -float(w[0]), -float(w[1]), -float(w[2]), -float(w[3]), int(a[0]), int(a[1]),
                                   dtype=dtype)                  # :125-128
            score_in[p].append(np.array(src, dtype))
            sdt[p].append(out); ix[p].append(x_); iy[p].append(y_)
        pp = par[p]
        for m in range(len(fid[pp])):                            # :134-156
            best = None
            for mm in range(K):
                wv = (sdt[p][mm] + T(model.biasw[bid[p][mm] + m])).astype(dtype)
                if K == 1:
                    best = wv
                elif best is None:
                    best = np.where(wv > T(-np.inf), wv, T(-np.inf)).astype(dtype)
                else:
                    best = np.where(wv > best, wv, best)
            f = fid[pp][m]
            base = acc.get(f, resp[f]).astype(dtype)
            acc[f] = (base + best).astype(dtype)
    K0 = len(fid[0])
    bias = T(model.biasw[bid[0][0]])                             # root.bias(0)[0] (:165)
    rv, ri = None, None
    for m in range(K0):
        wv = (acc.get(fid[0][m], resp[fid[0][m]]).astype(dtype) + bias).astype(dtype)
        if K0 == 1:
            rv, ri = wv, np.zeros(wv.shape, np.int32)
        elif rv is None:
            rv, ri = np.where(wv > T(-np.inf), wv, T(-np.inf)).astype(dtype), np.zeros(wv.shape, np.int32)
        else:
            take = wv > rv
            rv, ri = np.where(take, wv, rv), np.where(take, m, ri).astype(np.int32)
    return dict(score_in=score_in, sdt=sdt, ix=ix, iy=iy, rootv=rv, rooti=ri)


def backtrack(model, comp, maps, x, y):
    """argmin (:219-245) for one root location from level_maps' composed pointers -> locs [P, 3]."""
    P = model.nparts(comp)
    fid, bid, par = model.filterid[comp], model.biasid[comp], model.parentid[comp]
    locs = np.zeros((P, 3), np.int32)
    locs[0] = (x, y, maps["rooti"][y, x])
    T = maps["rootv"].dtype.type
    for p in range(1, P):
        px, py, pm = locs[par[p]]
        K = len(fid[p])
        best, bi = None, 0
        for mm in range(K):
            wv = T(maps["sdt"][p][mm][py, px] + T(model.biasw[bid[p][mm] + pm]))
            if best is None or wv > best:
                best, bi = wv, mm
        if K == 1:
            bi = 0
        locs[p] = (maps["ix"][p][bi][py, px], maps["iy"][p][bi][py, px], bi)
    return locs


def _subtree_size(par, p):
    n, P = 0, len(par)
    inside = [False] * P
    inside[p] = True
    for q in range(p, P):
        if q == p or (par[q] >= 0 and inside[par[q]]):
            inside[q] = True
            n += 1
    return n


def divergence_margin(model, comp, maps, locs_ref, locs_got):
    """First part (in index order; parents precede children) whose (x, y, mixture) differs while its parent's
    agree, and the gap between the two alternatives in the REFERENCE's numbers (`maps` from the reference-order
    responses).  The reference picks, at the parent location (px, py) with parent mixture pm:
      mm = argmax_mm sdt[mm](py,px) + bias(mm)[pm]                          (Math::reduceMax, :143)
      x  = argmax_n' in_mm[py, n'] + fx(px + ax - n')                       (x pass, DistanceTransform.hpp:216-218)
      y  = argmax_m' tmp_mm[m', x] + fy(py + ay - m')                       (y pass read at column x, :233-244)
    Returns (part, kind, margin >= 0, subtree size) or None when the configurations are equal."""
    par = model.parentid[comp]
    fid, did, bid = model.filterid[comp], model.defid[comp], model.biasid[comp]
    for p in range(1, model.nparts(comp)):
        if np.array_equal(locs_ref[p], locs_got[p]):
            continue
        if not np.array_equal(locs_ref[par[p]], locs_got[par[p]]):
            continue
        px, py, pm = (int(v) for v in locs_ref[par[p]])
        xo, yo, mo = (int(v) for v in locs_ref[p])
        xg, yg, mg = (int(v) for v in locs_got[p])
        sub = _subtree_size(par, p)
        if mo != mg:
            wo = float(maps["sdt"][p][mo][py, px]) + float(model.biasw[bid[p][mo] + pm])
            wg = float(maps["sdt"][p][mg][py, px]) + float(model.biasw[bid[p][mg] + pm])
            return p, "mixture", abs(wo - wg), sub
        w = model.defw[did[p][mo]].astype(np.float64)
        ax_, ay_ = (int(v) for v in model.anchors[did[p][mo]])
        src = maps["score_in"][p][mo].astype(np.float64)
        H, W = src.shape
        nn = np.arange(W, dtype=np.float64)
        if xo != xg:
            d = px + ax_ - nn
            obj = src[py] - w[0] * d * d - w[1] * d
            return p, "x", abs(float(obj[xo] - obj[xg])), sub
        # same column x: y-pass objective over the x-pass output of column x
        d = (xo + ax_) - nn[None, :]             # x pass at output column xo, every row
        tmp = (src - w[0] * d * d - w[1] * d).max(axis=1)
        mmv = np.arange(H, dtype=np.float64)
        dy = py + ay_ - mmv
        obj = tmp - w[2] * dy * dy - w[3] * dy
        return p, "y", abs(float(obj[yo] - obj[yg])), sub
    return None
