"""ctypes binding of libpbd_hip.so (include/pbd_c.h).

The library is the product; this module only marshals numpy arrays / device
pointers into it.  There is NO CPU fallback: if the shared object is missing
the import of `lib()` raises, and `pbd_create` fails on a box without a GPU.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .model import pbd_model_desc

_HERE = os.path.dirname(os.path.abspath(__file__))
# PBD_LIBRARY: load another build of the same ABI (tests/tools_*.py point it at libpbd_hip_probes.so, the
# `make probes` build with per-phase stamps and environment tuning knobs compiled in)
LIB_PATH = os.environ.get("PBD_LIBRARY") or os.path.join(_HERE, "libpbd_hip.so")

PBD_OK, PBD_ERR_ARG, PBD_ERR_UNSUPPORTED, PBD_ERR_CAPACITY, PBD_ERR_HIP, PBD_ERR_STATE, PBD_ERR_RCCL = range(7)
PBD_GATHER_AUTO, PBD_GATHER_HOST, PBD_GATHER_RCCL = 0, 1, 2
PBD_CONV_AUTO, PBD_CONV_EXACT, PBD_CONV_MFMA, PBD_CONV_SPLIT, PBD_CONV_SPLIT_F16 = 0, 1, 2, 3, 4
PBD_SCALAR_F32, PBD_SCALAR_F64 = 0, 1
PBD_DEPTH_8U, PBD_DEPTH_16U, PBD_DEPTH_32F, PBD_DEPTH_64F = 0, 2, 5, 6          # cv::Mat::depth() (src/HOGFeatures.cpp:136-146)
DEPTH_OF = {np.dtype(np.uint8): PBD_DEPTH_8U, np.dtype(np.uint16): PBD_DEPTH_16U, np.dtype(np.float32): PBD_DEPTH_32F,
            np.dtype(np.float64): PBD_DEPTH_64F}

EXPORTS = [
    "pbd_create", "pbd_destroy", "pbd_last_error", "pbd_max_parts", "pbd_set_stream",
    "pbd_detect_u8", "pbd_detect_dev_u8", "pbd_detect_enqueue_dev_u8", "pbd_detect_collect",
    "pbd_pyramid_geometry", "pbd_pyramid_u8", "pbd_get_level_image", "pbd_get_level_features",
    "pbd_set_level_features", "pbd_begin_frame", "pbd_pdf", "pbd_get_level_response",
    "pbd_set_level_response", "pbd_dp_min", "pbd_get_dp_pointers", "pbd_get_root", "pbd_dp_argmin",
    "pbd_dt2d", "pbd_hog_u8", "pbd_resize_u8", "pbd_pyrdown_u8", "pbd_nms_map",
    "pbd_set_levels", "pbd_get_level_features_f64", "pbd_set_level_features_f64", "pbd_get_level_response_f64",
    "pbd_set_level_response_f64", "pbd_get_root_f64", "pbd_dt2d_f64", "pbd_hog_u8_f64",
    "pbd_candidates_sort", "pbd_candidates_nms", "pbd_get_stage_ms", "pbd_set_profiling",
    "pbd_detect_enqueue_u8", "pbd_group_create", "pbd_group_destroy", "pbd_group_last_error", "pbd_group_size",
    "pbd_group_gather_mode", "pbd_group_member", "pbd_group_detect_batch_u8", "pbd_group_detect_u8",
    "pbd_get_work", "pbd_dp_timer", "pbd_debug_dt_stamps", "pbd_debug_hog_stamps", "pbd_debug_conv_stamps",
    "pbd_set_root", "pbd_set_root_f64", "pbd_set_dp_pointers", "pbd_get_footprint", "pbd_abi_version",
    "pbd_detect_batch_u8", "pbd_detect_batch_enqueue_u8", "pbd_detect_batch_enqueue_dev_u8", "pbd_detect_batch_collect",
    "pbd_get_stage_state", "pbd_get_conv_mode", "pbd_group_comm_size",
    "pbd_detect_image", "pbd_pyramid_image", "pbd_get_level_image_raw", "pbd_tune_plan",
]
PBD_ABI_VERSION = 4


class pbd_options(C.Structure):
    _fields_ = [("device", C.c_int32), ("conv_mode", C.c_int32), ("max_candidates", C.c_int32),
                ("dt_correct_ptr", C.c_int32), ("level_begin", C.c_int32), ("level_end", C.c_int32),
                ("scalar_type", C.c_int32), ("graph", C.c_int32), ("reserved", C.c_int32 * 2)]


class pbd_candidate_head(C.Structure):
    _fields_ = [("score", C.c_float), ("component", C.c_int32), ("level", C.c_int32), ("nparts", C.c_int32)]


HEAD_DTYPE = np.dtype([("score", np.float32), ("component", np.int32), ("level", np.int32), ("nparts", np.int32)])

_lib = None


class PbdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"pbd error {code}: {msg}")
        self.code = code


def lib() -> C.CDLL:
    """Load libpbd_hip.so; fail loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                              "g.build()'` (there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        L.pbd_last_error.restype = C.c_char_p
        L.pbd_last_error.argtypes = [C.c_void_p]
        L.pbd_group_last_error.restype = C.c_char_p
        L.pbd_group_last_error.argtypes = [C.c_void_p]
        L.pbd_group_member.restype = C.c_void_p
        L.pbd_group_member.argtypes = [C.c_void_p, C.c_int]
        for name in EXPORTS:
            getattr(L, name)  # every declared symbol must be exported
        if L.pbd_abi_version() != PBD_ABI_VERSION:
            raise ImportError(f"{LIB_PATH}: ABI version {L.pbd_abi_version()}, this binding is for {PBD_ABI_VERSION}")
        _lib = L
    return _lib


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


class Handle:
    """Owns one pbd_handle (one GPU, one stream)."""

    def __init__(self, model, device=0, conv_mode=PBD_CONV_AUTO, max_candidates=4096, dt_correct_ptr=0,
                 level_begin=0, level_end=0, dp_groups=0, dtype=np.float32, graph=0, dp_mode=0, nms_sz=0):
        """dtype: np.float32 = PartsBasedDetector<float>, np.float64 = PartsBasedDetector<double>.
        nms_sz > 0: score-map NMS of the root planes on the device in front of the back-tracking (pbd_options.reserved[0]).
        dp_mode: 0 = messages folded by the parent's x pass where the model allows it (default), 1 = the
        three-kernel structure (x pass, y pass, reduce) for every model.  dp_groups: ignored (kept for callers)."""
        self.L = lib()
        self.model = model
        self.desc = model.to_desc()
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise ValueError("dtype must be float32 or float64")
        self._f64 = self.dtype == np.dtype(np.float64)
        self._ct = C.c_double if self._f64 else C.c_float
        opt = pbd_options(device, conv_mode, max_candidates, dt_correct_ptr, level_begin, level_end,
                          PBD_SCALAR_F64 if self._f64 else PBD_SCALAR_F32, graph, (C.c_int32 * 2)(nms_sz, dp_mode))
        self.h = C.c_void_p()
        rc = self.L.pbd_create(C.byref(self.desc), C.byref(opt), C.byref(self.h))
        if rc != PBD_OK:
            msg = self.L.pbd_last_error(self.h).decode() if self.h else "allocation failed"
            if self.h:
                self.L.pbd_destroy(self.h)
            self.h = None
            raise PbdError(rc, msg)
        self.max_parts = self.L.pbd_max_parts(self.h)
        self.conv_mode = self.L.pbd_get_conv_mode(self.h)      # what PBD_CONV_AUTO resolved to

    def close(self):
        if getattr(self, "h", None):
            self.L.pbd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _fn(self, name):
        """Stage entry point of this handle's instantiation (name or name_f64)."""
        return getattr(self.L, name + ("_f64" if self._f64 else ""))

    def _chk(self, rc):
        if rc != PBD_OK:
            raise PbdError(rc, self.L.pbd_last_error(self.h).decode())

    # ---- detect ----------------------------------------------------------------
    def _bufs(self, capacity):
        heads = np.zeros(capacity, HEAD_DTYPE)
        boxes = np.zeros((capacity, self.max_parts, 4), np.int32)
        locs = np.zeros((capacity, self.max_parts, 3), np.int32)
        return heads, boxes, locs

    def _out(self, heads, boxes, locs, n):
        return heads[:n].copy(), boxes[:n].copy(), locs[:n].copy()

    def detect(self, im: np.ndarray, capacity=4096):
        im = np.ascontiguousarray(im, np.uint8)
        hgt, w = im.shape[:2]
        cn = 1 if im.ndim == 2 else im.shape[2]
        heads, boxes, locs = self._bufs(capacity)
        cnt = C.c_int(0)
        self._chk(self.L.pbd_detect_u8(self.h, _p(im, C.c_uint8), w, hgt, cn, w * cn,
                                       heads.ctypes.data_as(C.c_void_p), _p(boxes, C.c_int32), _p(locs, C.c_int32),
                                       capacity, C.byref(cnt)))
        return self._out(heads, boxes, locs, cnt.value)

    def detect_image(self, im: np.ndarray, capacity=4096):
        """pbd_detect_image: the image in its own depth (uint8 / uint16 / float32 / float64 = CV_8U / 16U / 32F / 64F); other dtypes are refused
        by the library the way the reference refuses them (CV_Error(StsUnsupportedFormat))"""
        im = np.ascontiguousarray(im)
        hgt, w = im.shape[:2]
        cn = 1 if im.ndim == 2 else im.shape[2]
        depth = DEPTH_OF.get(im.dtype, 1 if im.dtype == np.int8 else 3 if im.dtype == np.int16 else 4 if im.dtype == np.int32 else 7)
        heads, boxes, locs = self._bufs(capacity)
        cnt = C.c_int(0)
        self._chk(self.L.pbd_detect_image(self.h, im.ctypes.data_as(C.c_void_p), depth, w, hgt, cn, w * cn * im.itemsize,
                                          heads.ctypes.data_as(C.c_void_p), _p(boxes, C.c_int32), _p(locs, C.c_int32),
                                          capacity, C.byref(cnt)))
        return self._out(heads, boxes, locs, cnt.value)

    def tune_plan(self, im, batch=1):
        """pbd_tune_plan: (chosen geometry 1 / 2 — 0 for double handles —, [ms with 256 lanes / 40 KB, ms with 128 lanes / 25 KB]); im=None: the rule again"""
        chosen = C.c_int(0)
        ms = (C.c_double * 2)()
        if im is None:
            self._chk(self.L.pbd_tune_plan(self.h, None, 0, 0, 0, 0, 1, C.byref(chosen), ms))
            return 0, [0.0, 0.0]
        im = np.ascontiguousarray(im, np.uint8)
        hgt, w = im.shape[:2]
        cn = 1 if im.ndim == 2 else im.shape[2]
        self._chk(self.L.pbd_tune_plan(self.h, _p(im, C.c_uint8), w, hgt, cn, w * cn, int(batch), C.byref(chosen), ms))
        return chosen.value, [ms[0], ms[1]]

    def detect_dev(self, dptr: int, w, hgt, cn, stride=None, capacity=4096):
        heads, boxes, locs = self._bufs(capacity)
        cnt = C.c_int(0)
        self._chk(self.L.pbd_detect_dev_u8(self.h, C.c_void_p(dptr), w, hgt, cn, stride or w * cn,
                                           heads.ctypes.data_as(C.c_void_p), _p(boxes, C.c_int32),
                                           _p(locs, C.c_int32), capacity, C.byref(cnt)))
        return self._out(heads, boxes, locs, cnt.value)

    def enqueue_dev(self, dptr: int, w, hgt, cn, stride=None):
        self._chk(self.L.pbd_detect_enqueue_dev_u8(self.h, C.c_void_p(dptr), w, hgt, cn, stride or w * cn))

    def enqueue(self, im: np.ndarray):
        """pbd_detect_enqueue_u8: asynchronous H2D of a host image (pinned for true asynchrony) + all kernels.
        The array must stay alive until collect()."""
        hgt, w = im.shape[:2]
        cn = 1 if im.ndim == 2 else im.shape[2]
        assert im.dtype == np.uint8 and im.flags["C_CONTIGUOUS"]
        self._inflight_im = im
        self._chk(self.L.pbd_detect_enqueue_u8(self.h, C.c_void_p(im.ctypes.data), w, hgt, cn, w * cn))

    def enqueue_host_ptr(self, ptr: int, w, hgt, cn, stride=None):
        """same from a raw host pointer (e.g. a torch pinned tensor's data_ptr())."""
        self._chk(self.L.pbd_detect_enqueue_u8(self.h, C.c_void_p(ptr), w, hgt, cn, stride or w * cn))

    def collect(self, capacity=4096):
        heads, boxes, locs = self._bufs(capacity)
        cnt = C.c_int(0)
        self._chk(self.L.pbd_detect_collect(self.h, heads.ctypes.data_as(C.c_void_p), _p(boxes, C.c_int32),
                                            _p(locs, C.c_int32), capacity, C.byref(cnt)))
        return self._out(heads, boxes, locs, cnt.value)

    # ---- a batch of same-sized frames through this handle ------------------------------------
    def detect_batch(self, frames, capacity=4096):
        """pbd_detect_batch_u8: list of HxWxC uint8 frames -> list of (heads, boxes, locs), one per frame."""
        frames = [np.ascontiguousarray(f, np.uint8) for f in frames]
        if not frames:
            raise ValueError("detect_batch: at least one frame")
        if any(f.shape != frames[0].shape for f in frames):   # the C side reads w * hgt * cn bytes behind every pointer
            raise ValueError("detect_batch: all frames of a batch must have the same shape, got "
                             f"{sorted({f.shape for f in frames})}")
        hgt, w = frames[0].shape[:2]
        cn = 1 if frames[0].ndim == 2 else frames[0].shape[2]
        ptrs = (C.c_void_p * len(frames))(*[f.ctypes.data for f in frames])
        return self._batch_out(len(frames), capacity, lambda hd, bx, lc, cnt: self.L.pbd_detect_batch_u8(
            self.h, ptrs, len(frames), w, hgt, cn, w * cn, hd, bx, lc, capacity, cnt))

    def enqueue_batch_dev(self, dptr: int, nframes, w, hgt, cn):
        """frames back to back in device memory (tightly packed); collect with collect_batch"""
        self._nb = nframes
        self._chk(self.L.pbd_detect_batch_enqueue_dev_u8(self.h, C.c_void_p(dptr), nframes, w, hgt, cn))

    def enqueue_batch_host_ptrs(self, ptrs, w, hgt, cn):
        """host frames (raw pointers, e.g. pinned torch tensors); collect with collect_batch"""
        self._nb = len(ptrs)
        arr = (C.c_void_p * len(ptrs))(*ptrs)
        self._chk(self.L.pbd_detect_batch_enqueue_u8(self.h, arr, len(ptrs), w, hgt, cn, w * cn))

    def collect_batch(self, capacity=4096):
        return self._batch_out(self._nb, capacity, lambda hd, bx, lc, cnt: self.L.pbd_detect_batch_collect(self.h, hd, bx, lc, capacity, cnt))

    def _batch_out(self, nb, capacity, call):
        heads = np.zeros(nb * capacity, HEAD_DTYPE)
        boxes = np.zeros((nb * capacity, self.max_parts, 4), np.int32)
        locs = np.zeros((nb * capacity, self.max_parts, 3), np.int32)
        counts = (C.c_int * nb)()
        self._chk(call(heads.ctypes.data_as(C.c_void_p), _p(boxes, C.c_int32), _p(locs, C.c_int32), counts))
        return [self._out(heads[f * capacity:(f + 1) * capacity], boxes[f * capacity:(f + 1) * capacity],
                          locs[f * capacity:(f + 1) * capacity], counts[f]) for f in range(nb)]

    def set_levels(self, levels):
        """Process only this set of pyramid levels (empty = all): multi-GPU level sharding."""
        a = np.ascontiguousarray(list(levels), np.int32)
        self._chk(self.L.pbd_set_levels(self.h, _p(a, C.c_int32) if len(a) else None, len(a)))

    def set_stream(self, stream_ptr: int):
        self._chk(self.L.pbd_set_stream(self.h, C.c_void_p(stream_ptr)))

    # ---- stages ------------------------------------------------------------------
    def geometry(self, w, hgt):
        n = C.c_int(0)
        self._chk(self.L.pbd_pyramid_geometry(self.h, w, hgt, C.byref(n), None, None, None, None, None))
        a = [np.zeros(n.value, np.int32) for _ in range(4)]
        sc = np.zeros(n.value, np.float32)
        self._chk(self.L.pbd_pyramid_geometry(self.h, w, hgt, C.byref(n), *[_p(x, C.c_int32) for x in a],
                                              _p(sc, C.c_float)))
        return dict(nlevels=n.value, img_w=a[0], img_h=a[1], cell_w=a[2], cell_h=a[3], scales=sc)

    def begin_frame(self, w, hgt, cn):
        self._chk(self.L.pbd_begin_frame(self.h, w, hgt, cn))
        self._geo = self.geometry(w, hgt)
        self._cn = cn

    def pyramid(self, im: np.ndarray):
        im = np.ascontiguousarray(im, np.uint8)
        hgt, w = im.shape[:2]
        cn = 1 if im.ndim == 2 else im.shape[2]
        self._chk(self.L.pbd_pyramid_u8(self.h, _p(im, C.c_uint8), w, hgt, cn, w * cn))
        self._geo = self.geometry(w, hgt)
        self._cn = cn
        self._imdtype = np.dtype(np.uint8)

    def pyramid_image(self, im: np.ndarray):
        """pbd_pyramid_image: pyramid() for an image of any accepted depth (its numpy dtype)"""
        im = np.ascontiguousarray(im)
        hgt, w = im.shape[:2]
        cn = 1 if im.ndim == 2 else im.shape[2]
        self._chk(self.L.pbd_pyramid_image(self.h, im.ctypes.data_as(C.c_void_p), DEPTH_OF[im.dtype], w, hgt, cn, w * cn * im.itemsize))
        self._geo = self.geometry(w, hgt)
        self._cn = cn
        self._imdtype = im.dtype

    def level_image_raw(self, l):
        g = self._geo
        shape = (g["img_h"][l], g["img_w"][l]) + ((self._cn,) if self._cn > 1 else ())
        out = np.zeros(shape, getattr(self, "_imdtype", np.dtype(np.uint8)))
        self._chk(self.L.pbd_get_level_image_raw(self.h, l, out.ctypes.data_as(C.c_void_p), C.c_size_t(out.nbytes)))
        return out

    def level_image(self, l):
        g = self._geo
        shape = (g["img_h"][l], g["img_w"][l]) + ((self._cn,) if self._cn > 1 else ())
        out = np.zeros(shape, np.uint8)
        self._chk(self.L.pbd_get_level_image(self.h, l, _p(out, C.c_uint8)))
        return out

    def level_features(self, l):
        g = self._geo
        out = np.zeros((g["cell_h"][l], g["cell_w"][l], 32), self.dtype)
        self._chk(self._fn("pbd_get_level_features")(self.h, l, _p(out, self._ct)))
        return out

    def set_level_features(self, l, f):
        f = np.ascontiguousarray(f, self.dtype)
        self._chk(self._fn("pbd_set_level_features")(self.h, l, _p(f, self._ct)))

    def pdf(self):
        self._chk(self.L.pbd_pdf(self.h))

    def level_response(self, l, n):
        g = self._geo
        out = np.zeros((g["cell_h"][l], g["cell_w"][l]), self.dtype)
        self._chk(self._fn("pbd_get_level_response")(self.h, l, n, _p(out, self._ct)))
        return out

    def set_level_response(self, l, n, r):
        r = np.ascontiguousarray(r, self.dtype)
        self._chk(self._fn("pbd_set_level_response")(self.h, l, n, _p(r, self._ct)))

    def dp_min(self):
        self._chk(self.L.pbd_dp_min(self.h))

    def stage_state(self):
        """pbd_get_stage_state: dict of which stage buffers currently hold valid data"""
        st = (C.c_int32 * 4)()
        self._chk(self.L.pbd_get_stage_state(self.h, st))
        return dict(pyramid=bool(st[0]), features=bool(st[1]), responses=bool(st[2]), dp=bool(st[3]))

    def dp_pointers(self, l, c, p, m):
        g = self._geo
        sh = (g["cell_h"][l], g["cell_w"][l])
        ix, iy, ik = (np.zeros(sh, np.int32) for _ in range(3))
        self._chk(self.L.pbd_get_dp_pointers(self.h, l, c, p, m, _p(ix, C.c_int32), _p(iy, C.c_int32),
                                             _p(ik, C.c_int32)))
        return ix, iy, ik

    def root(self, l, c):
        g = self._geo
        sh = (g["cell_h"][l], g["cell_w"][l])
        rv, ri = np.zeros(sh, self.dtype), np.zeros(sh, np.int32)
        self._chk(self._fn("pbd_get_root")(self.h, l, c, _p(rv, self._ct), _p(ri, C.c_int32)))
        return rv, ri

    def set_root(self, l, c, rootv, rooti):
        rv, ri = np.ascontiguousarray(rootv, self.dtype), np.ascontiguousarray(rooti, np.int32)
        self._chk(self._fn("pbd_set_root")(self.h, l, c, _p(rv, self._ct), _p(ri, C.c_int32)))

    def set_dp_pointers(self, l, c, p, m, ix, iy, ik):
        """hand DynamicProgram::argmin pointer tables that this handle's min() did not produce"""
        a = [np.ascontiguousarray(t, np.int32) for t in (ix, iy, ik)]
        self._chk(self.L.pbd_set_dp_pointers(self.h, l, c, p, m, *[_p(t, C.c_int32) for t in a]))

    def footprint(self):
        """(frame_bytes, model_bytes) of device memory held by the handle"""
        fb, mb = C.c_size_t(0), C.c_size_t(0)
        self._chk(self.L.pbd_get_footprint(self.h, C.byref(fb), C.byref(mb)))
        return fb.value, mb.value

    def dp_argmin(self, capacity=4096):
        heads, boxes, locs = self._bufs(capacity)
        cnt = C.c_int(0)
        self._chk(self.L.pbd_dp_argmin(self.h, heads.ctypes.data_as(C.c_void_p), _p(boxes, C.c_int32),
                                       _p(locs, C.c_int32), capacity, C.byref(cnt)))
        return self._out(heads, boxes, locs, cnt.value)

    # ---- primitives ----------------------------------------------------------------
    def dt2d(self, a: np.ndarray, ax, bx, ay, by, osx, osy):
        a = np.ascontiguousarray(a, self.dtype)
        out = np.zeros_like(a)
        ix, iy = np.zeros(a.shape, np.int32), np.zeros(a.shape, np.int32)
        self._chk(self._fn("pbd_dt2d")(self.h, _p(a, self._ct), a.shape[0], a.shape[1], C.c_double(ax), C.c_double(bx),
                                       C.c_double(ay), C.c_double(by), osx, osy, _p(out, self._ct), _p(ix, C.c_int32),
                                       _p(iy, C.c_int32)))
        return out, ix, iy

    def hog(self, im: np.ndarray):
        im = np.ascontiguousarray(im, np.uint8)
        hgt, w = im.shape[:2]
        cn = 1 if im.ndim == 2 else im.shape[2]
        sb = self.model.sbin
        buf = np.zeros((hgt // sb + 2) * (w // sb + 2) * 32, self.dtype)
        a, b = C.c_int(0), C.c_int(0)
        self._chk(self._fn("pbd_hog_u8")(self.h, _p(im, C.c_uint8), w, hgt, cn, w * cn, _p(buf, self._ct), C.byref(a),
                                         C.byref(b)))
        return buf[: b.value * a.value * 32].reshape(b.value, a.value, 32).copy()

    def resize(self, im: np.ndarray, ow, oh):
        im = np.ascontiguousarray(im, np.uint8)
        hgt, w = im.shape[:2]
        cn = 1 if im.ndim == 2 else im.shape[2]
        out = np.zeros((oh, ow) + ((cn,) if cn > 1 else ()), np.uint8)
        self._chk(self.L.pbd_resize_u8(self.h, _p(im, C.c_uint8), w, hgt, cn, w * cn, _p(out, C.c_uint8), ow, oh))
        return out

    def pyrdown(self, im: np.ndarray):
        im = np.ascontiguousarray(im, np.uint8)
        hgt, w = im.shape[:2]
        cn = 1 if im.ndim == 2 else im.shape[2]
        out = np.zeros(((hgt + 1) // 2, (w + 1) // 2) + ((cn,) if cn > 1 else ()), np.uint8)
        self._chk(self.L.pbd_pyrdown_u8(self.h, _p(im, C.c_uint8), w, hgt, cn, w * cn, _p(out, C.c_uint8)))
        return out

    def nms_map(self, src: np.ndarray, sz: int):
        src = np.ascontiguousarray(src, np.float32)
        out = np.zeros(src.shape, np.uint8)
        self._chk(self.L.pbd_nms_map(self.h, _p(src, C.c_float), src.shape[0], src.shape[1], sz, _p(out, C.c_uint8)))
        return out

    # ---- instrumentation -------------------------------------------------------------
    def set_profiling(self, on=True):
        self._chk(self.L.pbd_set_profiling(self.h, int(on)))

    def stage_ms(self):
        ms = (C.c_float * 6)()
        self._chk(self.L.pbd_get_stage_ms(self.h, ms))
        return dict(zip(["image_pyramid", "hog", "pdf", "dp_min", "argmin", "total"], list(ms)))

    def work(self):
        wk = (C.c_double * 6)()
        self._chk(self.L.pbd_get_work(self.h, wk))
        return dict(zip(["B_hog", "B_pdf", "F_pdf", "B_dp", "cells", "dt_elements"], list(wk)))

    def dp_timer(self, reset=False):
        ms, n = C.c_double(0), C.c_int(0)
        self._chk(self.L.pbd_dp_timer(self.h, int(reset), C.byref(ms), C.byref(n)))
        return ms.value, n.value


class Group:
    """pbd_group: one process driving several GPUs (include/pbd_c.h).  devices may repeat an ordinal."""

    def __init__(self, model, devices, gather=PBD_GATHER_AUTO, conv_mode=PBD_CONV_AUTO, max_candidates=4096,
                 dtype=np.float32, graph=0, nms_sz=0):
        self.L = lib()
        self.model = model
        self.desc = model.to_desc()
        f64 = np.dtype(dtype) == np.dtype(np.float64)
        opt = pbd_options(0, conv_mode, max_candidates, 0, 0, 0, PBD_SCALAR_F64 if f64 else PBD_SCALAR_F32, graph,
                          (C.c_int32 * 2)(int(nms_sz), 0))
        dv = np.ascontiguousarray(list(devices), np.int32)
        self.g = C.c_void_p()
        rc = self.L.pbd_group_create(C.byref(self.desc), C.byref(opt), _p(dv, C.c_int32), len(dv), gather, C.byref(self.g))
        if rc != PBD_OK:
            msg = self.L.pbd_group_last_error(self.g).decode() if self.g else "allocation failed"
            if self.g:
                self.L.pbd_group_destroy(self.g)
            self.g = None
            raise PbdError(rc, msg)
        self.size = self.L.pbd_group_size(self.g)
        self.gather_mode = self.L.pbd_group_gather_mode(self.g)
        self.comm_size = self.L.pbd_group_comm_size(self.g)       # ranks of the RCCL communicator (0: host gather)
        self.max_parts = self.L.pbd_max_parts(C.c_void_p(self.L.pbd_group_member(self.g, 0)))

    def close(self):
        if getattr(self, "g", None):
            self.L.pbd_group_destroy(self.g)
            self.g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != PBD_OK:
            raise PbdError(rc, self.L.pbd_group_last_error(self.g).decode())

    def detect_batch(self, frames, capacity=4096):
        """frames: list of equal-sized uint8 images -> list of (heads, boxes, locs), frame f on member f % size."""
        frames = [np.ascontiguousarray(f, np.uint8) for f in frames]
        n = len(frames)
        if not n or any(f.shape != frames[0].shape for f in frames):
            raise ValueError("detect_batch: one or more frames, all of the same shape")
        hgt, w = frames[0].shape[:2]
        cn = 1 if frames[0].ndim == 2 else frames[0].shape[2]
        ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
        heads = np.zeros(n * capacity, HEAD_DTYPE)
        boxes = np.zeros((n * capacity, self.max_parts, 4), np.int32)
        locs = np.zeros((n * capacity, self.max_parts, 3), np.int32)
        counts = np.zeros(n, np.int32)
        self._chk(self.L.pbd_group_detect_batch_u8(self.g, ptrs, n, w, hgt, cn, w * cn, heads.ctypes.data_as(C.c_void_p),
                                                   _p(boxes, C.c_int32), _p(locs, C.c_int32), capacity, _p(counts, C.c_int32)))
        return [(heads[f * capacity: f * capacity + counts[f]].copy(), boxes[f * capacity: f * capacity + counts[f]].copy(),
                 locs[f * capacity: f * capacity + counts[f]].copy()) for f in range(n)]

    def detect(self, im, capacity=4096):
        """one frame, pyramid levels LPT-sharded over the members."""
        im = np.ascontiguousarray(im, np.uint8)
        hgt, w = im.shape[:2]
        cn = 1 if im.ndim == 2 else im.shape[2]
        heads = np.zeros(capacity, HEAD_DTYPE)
        boxes = np.zeros((capacity, self.max_parts, 4), np.int32)
        locs = np.zeros((capacity, self.max_parts, 3), np.int32)
        cnt = C.c_int(0)
        self._chk(self.L.pbd_group_detect_u8(self.g, _p(im, C.c_uint8), w, hgt, cn, w * cn, heads.ctypes.data_as(C.c_void_p),
                                             _p(boxes, C.c_int32), _p(locs, C.c_int32), capacity, C.byref(cnt)))
        return heads[:cnt.value].copy(), boxes[:cnt.value].copy(), locs[:cnt.value].copy()


def candidates_sort(heads, boxes, locs):
    """Candidate::sort (include/Candidate.hpp:91-99) — host code inside the library."""
    heads, boxes, locs = heads.copy(), np.ascontiguousarray(boxes).copy(), np.ascontiguousarray(locs).copy()
    mp = boxes.shape[1] if boxes.ndim == 3 else 1
    rc = lib().pbd_candidates_sort(heads.ctypes.data_as(C.c_void_p), _p(boxes, C.c_int32), _p(locs, C.c_int32),
                                   len(heads), mp)
    if rc:
        raise PbdError(rc, "pbd_candidates_sort")
    return heads, boxes, locs


def candidates_nms(heads, boxes, locs, im_w, im_h, overlap=0.0):
    """Candidate::nonMaximaSuppression (include/Candidate.hpp:277-304)."""
    heads, boxes, locs = heads.copy(), np.ascontiguousarray(boxes).copy(), np.ascontiguousarray(locs).copy()
    mp = boxes.shape[1]
    kept = C.c_int(0)
    rc = lib().pbd_candidates_nms(heads.ctypes.data_as(C.c_void_p), _p(boxes, C.c_int32), _p(locs, C.c_int32),
                                  len(heads), mp, im_w, im_h, C.c_float(overlap), C.byref(kept))
    if rc:
        raise PbdError(rc, "pbd_candidates_nms")
    return heads[:kept.value], boxes[:kept.value], locs[:kept.value]
