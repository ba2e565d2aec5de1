"""ctypes binding of oracle/liborc.so (pbd_oracle.c).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Never by partsbaseddetector_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "liborc.so")
HEAD_DTYPE = np.dtype([("score", np.float32), ("component", np.int32), ("level", np.int32), ("nparts", np.int32)])
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        for sfx in ("", "_f64"):
            for n in ("orc_frame_image", "orc_frame_feat", "orc_frame_resp", "orc_frame_ix", "orc_frame_iy",
                      "orc_frame_ik", "orc_frame_rootv", "orc_frame_rooti"):
                getattr(L, n + sfx).restype = C.c_void_p
                getattr(L, n + sfx).argtypes = [C.c_void_p, C.c_int]
            getattr(L, "orc_frame_free" + sfx).argtypes = [C.c_void_p]
            getattr(L, "orc_frame_nlevels" + sfx).argtypes = [C.c_void_p]
            getattr(L, "orc_frame_dims" + sfx).argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _sfx(dtype):
    """Reference instantiation: np.float32 -> PartsBasedDetector<float> (orc_*), np.float64 -> <double> (orc_*_f64)."""
    dt = np.dtype(dtype)
    if dt == np.float32:
        return ""
    if dt == np.float64:
        return "_f64"
    raise ValueError("dtype must be float32 or float64")


def _fn(name, dtype):
    return getattr(lib(), name + _sfx(dtype))


def geometry(w, h, sbin, interval):
    n = C.c_int(0)
    a = [np.zeros(128, np.int32) for _ in range(4)]
    sc = np.zeros(128, np.float32)
    rc = lib().orc_pyramid_geometry(w, h, sbin, interval, C.byref(n), *[_p(x) for x in a], _p(sc))
    if rc:
        raise ValueError("image too small for the pyramid")
    k = n.value
    return dict(nlevels=k, img_w=a[0][:k], img_h=a[1][:k], cell_w=a[2][:k], cell_h=a[3][:k], scales=sc[:k])


def _cn(im):
    return 1 if im.ndim == 2 else im.shape[2]


# image depths the reference dispatches on (src/HOGFeatures.cpp:136-146): numpy dtype -> (cv::Mat::depth() code, oracle suffix)
DEPTHS = {np.dtype(np.uint8): (0, "8u"), np.dtype(np.uint16): (2, "16u"), np.dtype(np.float32): (5, "32f"), np.dtype(np.float64): (6, "64f")}


def _img(im):
    """an image as the reference would receive it: uint8 unless it already has one of the other accepted depths"""
    im = np.asarray(im)
    if im.dtype not in DEPTHS:
        im = im.astype(np.uint8)
    return np.ascontiguousarray(im)


def resize(im, ow, oh):
    im = _img(im)
    h, w = im.shape[:2]
    cn = _cn(im)
    out = np.zeros((oh, ow) + ((cn,) if cn > 1 else ()), im.dtype)
    getattr(lib(), "orc_resize_linear_" + DEPTHS[im.dtype][1])(_p(im), w, h, cn, w * cn, _p(out), ow, oh)   # (strides in elements)
    return out


def pyrdown(im):
    im = _img(im)
    h, w = im.shape[:2]
    cn = _cn(im)
    out = np.zeros(((h + 1) // 2, (w + 1) // 2) + ((cn,) if cn > 1 else ()), im.dtype)
    getattr(lib(), "orc_pyrdown_" + DEPTHS[im.dtype][1])(_p(im), w, h, cn, w * cn, _p(out))
    return out


def hog(im, sbin, dtype=np.float32):
    im = _img(im)
    h, w = im.shape[:2]
    cn = _cn(im)
    cw, ch = C.c_int(0), C.c_int(0)
    lib().orc_cells_of(w, h, sbin, C.byref(cw), C.byref(ch))
    out = np.zeros((ch.value, cw.value, 32), dtype)
    rc = _fn("orc_hog_image", dtype)(_p(im), DEPTHS[im.dtype][0], w, h, cn, w * cn * im.itemsize, sbin, _p(out))
    assert rc == 0
    return out


def pdf_level(feat, filters, dtype=np.float32):
    """feat [H, W, 32]; filters list of kh x (kw*32) -> [nf, H, W]."""
    feat = np.ascontiguousarray(feat, dtype)
    H, W, flen = feat.shape
    filt = np.ascontiguousarray(np.stack(filters).astype(np.float32))
    nf, kh = filt.shape[0], filt.shape[1]
    kw = filt.shape[2] // flen
    out = np.zeros((nf, H, W), dtype)
    _fn("orc_pdf_level", dtype)(_p(feat), H, W, flen, _p(filt), nf, kh, kw, _p(out))
    return out


def dt1d(src, a, b, os_, dtype=np.float32):
    src = np.ascontiguousarray(src, dtype)
    dst = np.zeros_like(src)
    ptr = np.zeros(src.shape, np.int32)
    _fn("orc_dt1d", dtype)(_p(src), _p(dst), _p(ptr), src.shape[0], C.c_double(a), C.c_double(b), os_)
    return dst, ptr


def dt2d(a, ax, bx, ay, by, osx, osy, correct_ptr=0, dtype=np.float32):
    a = np.ascontiguousarray(a, dtype)
    out = np.zeros_like(a)
    ix, iy = np.zeros(a.shape, np.int32), np.zeros(a.shape, np.int32)
    _fn("orc_dt2d", dtype)(_p(a), a.shape[0], a.shape[1], C.c_double(ax), C.c_double(bx), C.c_double(ay), C.c_double(by),
                   osx, osy, _p(out), _p(ix), _p(iy), correct_ptr)
    return out, ix, iy


def ptr_planes(desc, comp):
    return lib().orc_ptr_planes(C.byref(desc), comp)


def dp_min_level(desc, comp, resp, correct_ptr=0, dtype=np.float32):
    """resp [nf, H, W] -> Ix, Iy, Ik [planes, H, W], rootv, rooti [H, W]."""
    resp = np.ascontiguousarray(resp, dtype)
    _, H, W = resp.shape
    npl = ptr_planes(desc, comp)
    Ix, Iy, Ik = (np.zeros((npl, H, W), np.int32) for _ in range(3))
    rv, ri = np.zeros((H, W), dtype), np.zeros((H, W), np.int32)
    _fn("orc_dp_min_level", dtype)(C.byref(desc), comp, _p(resp), H, W, _p(Ix), _p(Iy), _p(Ik), _p(rv), _p(ri), correct_ptr)
    return Ix, Iy, Ik, rv, ri


def dp_argmin_level(desc, comp, level, scale, rootv, rooti, Ix, Iy, Ik, capacity=8192, dtype=np.float32):
    """DynamicProgram<T>::argmin for one (level, component) on GIVEN tables (src/DynamicProgram.cpp:189-255):
    rootv/rooti [H, W], Ix/Iy/Ik [planes, H, W] -> (heads, boxes, locs) in row-major root order."""
    rootv = np.ascontiguousarray(rootv, dtype)
    rooti, Ix, Iy, Ik = (np.ascontiguousarray(a, np.int32) for a in (rooti, Ix, Iy, Ik))
    H, W = rootv.shape
    mp = lib().orc_max_parts(C.byref(desc))
    heads = np.zeros(capacity, HEAD_DTYPE)
    boxes = np.zeros((capacity, mp, 4), np.int32)
    locs = np.zeros((capacity, mp, 3), np.int32)
    cnt = C.c_int(0)
    _fn("orc_dp_argmin_level", dtype)(C.byref(desc), comp, level, C.c_float(scale), _p(rootv), _p(rooti), _p(Ix), _p(Iy), _p(Ik),
                                      H, W, mp, _p(heads), _p(boxes), _p(locs), capacity, C.byref(cnt))
    n = min(cnt.value, capacity)
    return heads[:n].copy(), boxes[:n].copy(), locs[:n].copy()


class Frame:
    """Intermediates of one orc_detect_u8 run."""

    def __init__(self, ptr, model, dtype=np.float32):
        self.ptr, self.model, self.dtype = ptr, model, np.dtype(dtype)
        self.nlevels = _fn("orc_frame_nlevels", dtype)(ptr)
        self.dims = []
        for l in range(self.nlevels):
            v = [C.c_int(0) for _ in range(4)]
            s = C.c_float(0)
            _fn("orc_frame_dims", dtype)(ptr, l, *[C.addressof(x) for x in v], C.addressof(s))
            self.dims.append(tuple(x.value for x in v) + (s.value,))

    def _arr(self, fn, l, shape, dtype):
        p = _fn(fn, self.dtype)(self.ptr, l)
        n = int(np.prod(shape))
        if n == 0:
            return np.zeros(shape, dtype)
        buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(p)
        return np.frombuffer(buf, dtype=dtype).reshape(shape).copy()

    def image(self, l, cn, dtype=np.uint8):
        iw, ih = self.dims[l][0], self.dims[l][1]
        return self._arr("orc_frame_image", l, (ih, iw) + ((cn,) if cn > 1 else ()), dtype)

    def feat(self, l):
        return self._arr("orc_frame_feat", l, (self.dims[l][3], self.dims[l][2], 32), self.dtype)

    def resp(self, l):
        return self._arr("orc_frame_resp", l, (len(self.model.filtersw), self.dims[l][3], self.dims[l][2]), self.dtype)

    def pointers(self, l, total_planes):
        sh = (total_planes, self.dims[l][3], self.dims[l][2])
        return tuple(self._arr(f, l, sh, np.int32) for f in ("orc_frame_ix", "orc_frame_iy", "orc_frame_ik"))

    def root(self, l):
        sh = (self.model.ncomponents, self.dims[l][3], self.dims[l][2])
        return self._arr("orc_frame_rootv", l, sh, self.dtype), self._arr("orc_frame_rooti", l, sh, np.int32)

    def free(self):
        if self.ptr:
            _fn("orc_frame_free", self.dtype)(self.ptr)
            self.ptr = None

    def __del__(self):
        self.free()


def detect(model, im, capacity=8192, keep=False, correct_ptr=0, desc=None, dtype=np.float32):
    """orc_detect_image[_f64] -> (heads, boxes, locs, stage_ms[, Frame]).  The image's own dtype is its depth (uint8, uint16, float32,
    float64: src/HOGFeatures.cpp:136-146); anything else is taken as uint8."""
    im = _img(im)
    h, w = im.shape[:2]
    cn = _cn(im)
    desc = desc or model.to_desc()
    mp = lib().orc_max_parts(C.byref(desc))
    heads = np.zeros(capacity, HEAD_DTYPE)
    boxes = np.zeros((capacity, mp, 4), np.int32)
    locs = np.zeros((capacity, mp, 3), np.int32)
    cnt = C.c_int(0)
    ms = (C.c_double * 5)()
    fp = C.c_void_p()
    rc = _fn("orc_detect_image", dtype)(C.byref(desc), _p(im), DEPTHS[im.dtype][0], w, h, cn, w * cn * im.itemsize, _p(heads), _p(boxes), _p(locs),
                                        capacity, C.byref(cnt), ms, C.byref(fp) if keep else None, correct_ptr)
    if rc:
        raise ValueError("orc_detect_image failed (image too small?)")
    n = min(cnt.value, capacity)
    res = (heads[:n].copy(), boxes[:n].copy(), locs[:n].copy(), list(ms))
    if keep:
        return res + (Frame(fp, model, dtype),)
    return res


def candidates_sort(heads, boxes, locs):
    heads, boxes, locs = heads.copy(), boxes.copy(), locs.copy()
    lib().orc_candidates_sort(_p(heads), _p(boxes), _p(locs), len(heads), boxes.shape[1])
    return heads, boxes, locs


def candidates_nms(heads, boxes, locs, im_w, im_h, overlap):
    heads, boxes, locs = heads.copy(), boxes.copy(), locs.copy()
    kept = C.c_int(0)
    lib().orc_candidates_nms(_p(heads), _p(boxes), _p(locs), len(heads), boxes.shape[1], im_w, im_h,
                             C.c_float(overlap), C.byref(kept))
    return heads[:kept.value], boxes[:kept.value], locs[:kept.value]


def nms_map(src, sz):
    src = np.ascontiguousarray(src, np.float32)
    out = np.zeros(src.shape, np.uint8)
    lib().orc_nms_map(_p(src), src.shape[0], src.shape[1], sz, _p(out))
    return out


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n: int):
    """omp_set_num_threads for the timed CPU baseline legs (all cores / one thread)."""
    lib().orc_set_num_threads(int(n))
