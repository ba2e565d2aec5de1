"""Host-side mirror of the reference's detector interfaces over the C ABI.

Same names, call order and argument meaning as the reference so that callers
(and the parity tests) read like code written against it:

    reference (C++)                                   here
    ------------------------------------------------  --------------------------------
    PartsBasedDetector<T>::distributeModel(Model&)    PartsBasedDetector.distributeModel(model)
    PartsBasedDetector<T>::detect(im, candidates)     PartsBasedDetector.detect(im) -> [Candidate]
    IFeatures::{binsize,nscales,scales,pyramid}       HOGFeatures.*
    IConvolutionEngine::{setFilters,pdf}              SpatialConvolutionEngine.*
    DynamicProgram<T>::{min,argmin}                   DynamicProgram.*
    Candidate::{sort,nonMaximaSuppression}            Candidate.*

(src/PartsBasedDetector.cpp:69-127, include/IFeatures.hpp:49-73,
include/IConvolutionEngine.hpp:44-68, include/DynamicProgram.hpp:61-77,
include/Candidate.hpp:56-111,277-304.)  All numerics run in libpbd_hip.so.
The three stage objects share one device handle: features, responses and DP
tables stay resident in HBM between the calls, exactly like the fused detect().
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from . import capi
from .model import Model


@dataclass
class Candidate:
    """include/Candidate.hpp:56-111: part boxes (x, y, w, h), part confidences, component."""

    parts: np.ndarray        # [nparts, 4] cv::Rect per part
    confidence: np.ndarray   # [nparts]; root = score, others 0 (src/DynamicProgram.cpp:241-244)
    component: int
    level: int = -1
    locs: Optional[np.ndarray] = None  # [nparts, 3] (x, y, mixture) in cells of `level`

    def score(self) -> float:
        return float(self.confidence[0]) if len(self.confidence) else float("-inf")

    def setScore(self, confidence: float) -> None:
        """Candidate::setScore (:76)."""
        if len(self.confidence) == 0:
            self.confidence = np.zeros(1, np.float32)
        self.confidence[0] = np.float32(confidence)

    def resize(self, factor: float) -> None:
        """Candidate::resize (:82-89): `int *= float` — float product truncated toward zero, per field."""
        f = np.float32(factor)
        self.parts = np.trunc(self.parts.astype(np.float32) * f).astype(self.parts.dtype)

    def boundingBox(self):
        x, y, w, h = [int(v) for v in self.parts[0]]
        for q in self.parts:
            x1, y1 = min(x, int(q[0])), min(y, int(q[1]))
            w = max(x + w, int(q[0] + q[2])) - x1
            h = max(y + h, int(q[1] + q[3])) - y1
            x, y = x1, y1
        return x, y, w, h

    @staticmethod
    def _pack(cands: List["Candidate"]):
        mp = max((len(c.parts) for c in cands), default=1)
        heads = np.zeros(len(cands), capi.HEAD_DTYPE)
        boxes = np.zeros((len(cands), mp, 4), np.int32)
        locs = np.zeros((len(cands), mp, 3), np.int32)
        for i, c in enumerate(cands):
            heads[i] = (c.score(), c.component, c.level, len(c.parts))
            boxes[i, : len(c.parts)] = c.parts
            if c.locs is not None:
                locs[i, : len(c.parts)] = c.locs
        return heads, boxes, locs

    @staticmethod
    def _unpack(heads, boxes, locs) -> List["Candidate"]:
        out = []
        for i in range(len(heads)):
            n = int(heads[i]["nparts"])
            conf = np.zeros(n, np.float32)
            conf[0] = heads[i]["score"]
            out.append(Candidate(boxes[i, :n].copy(), conf, int(heads[i]["component"]), int(heads[i]["level"]),
                                 locs[i, :n].copy()))
        return out

    @staticmethod
    def sort(candidates: List["Candidate"]) -> List["Candidate"]:
        """Candidate::sort — descending root score."""
        return Candidate._unpack(*capi.candidates_sort(*Candidate._pack(candidates)))

    @staticmethod
    def nonMaximaSuppression(im_shape, candidates: List["Candidate"], overlap: float = 0.0) -> List["Candidate"]:
        """Candidate::nonMaximaSuppression(im, candidates, overlap)."""
        h, w = im_shape[:2]
        return Candidate._unpack(*capi.candidates_nms(*Candidate._pack(candidates), w, h, overlap))


class HOGFeatures:
    """IFeatures implementation (include/HOGFeatures.hpp:52-88) on the device."""

    def __init__(self, handle: capi.Handle):
        self._h = handle
        self._scales = np.zeros(0, np.float32)
        self._nscales = handle.model.interval

    def binsize(self) -> int:
        return self._h.model.sbin

    def nscales(self) -> int:
        return self._nscales

    def scales(self) -> np.ndarray:
        return self._scales

    def pyramid(self, im: np.ndarray) -> List[np.ndarray]:
        """HOGFeatures<T>::pyramid: returns the feature pyramid (fine to coarse),
        each level H x (W*flen) like the reference's cv::Mat; it also stays resident.  The image's dtype is its depth (:136-146:
        uint8, uint16, float32, float64; anything else raises like CV_Error(StsUnsupportedFormat))."""
        if np.asarray(im).dtype == np.uint8:
            self._h.pyramid(im)
        elif np.asarray(im).dtype in capi.DEPTH_OF:
            self._h.pyramid_image(im)
        else:
            raise capi.PbdError(capi.PBD_ERR_UNSUPPORTED, "Unsupported image type")
        g = self._h._geo
        self._nscales, self._scales = g["nlevels"], g["scales"]
        return [self._h.level_features(l).reshape(g["cell_h"][l], -1) for l in range(g["nlevels"])]


class SpatialConvolutionEngine:
    """IConvolutionEngine implementation (include/SpatialConvolutionEngine.hpp:44-58)."""

    def __init__(self, handle: capi.Handle):
        self._h = handle

    def setFilters(self, filters) -> None:
        """Filters are uploaded (transposed for the kernels) at distributeModel time;
        like the reference this must precede pdf()."""
        if len(filters) != len(self._h.model.filtersw):
            raise ValueError("setFilters: filter bank differs from the distributed model")

    def pdf(self, features: Optional[List[np.ndarray]] = None) -> List[List[np.ndarray]]:
        """pdf(features, responses): responses[level][filter].  `features=None`
        uses the pyramid already resident on the device."""
        g = self._h._geo
        if features is not None:
            for l, f in enumerate(features):
                # the handle's T (float or double): a double detector must not round injected features to float
                self._h.set_level_features(l, np.asarray(f, self._h.dtype).reshape(g["cell_h"][l], g["cell_w"][l], 32))
        self._h.pdf()
        nf = len(self._h.model.filtersw)
        return [[self._h.level_response(l, n) for n in range(nf)] for l in range(g["nlevels"])]


class DynamicProgram:
    """DynamicProgram<T> (include/DynamicProgram.hpp:61-77)."""

    def __init__(self, handle: capi.Handle):
        self._h = handle

    def min(self, scores: Optional[List[List[np.ndarray]]] = None):
        """min(parts, scores, Ix, Iy, Ik, rootv, rooti): returns (Ix, Iy, Ik, rootv, rooti)
        indexed [level][component][part][parent mixture] / [level][component]."""
        h, g, m = self._h, self._h._geo, self._h.model
        if scores is not None:
            for l in range(g["nlevels"]):
                for n, r in enumerate(scores[l]):
                    h.set_level_response(l, n, r)
        h.dp_min()
        Ix, Iy, Ik, rootv, rooti = [], [], [], [], []
        for l in range(g["nlevels"]):
            lx, ly, lk, rv, ri = [], [], [], [], []
            for c in range(m.ncomponents):
                cx, cy, ck = [[]], [[]], [[]]
                for p in range(1, m.nparts(c)):
                    L = len(m.filterid[c][m.parentid[c][p]])
                    trip = [h.dp_pointers(l, c, p, pm) for pm in range(L)]
                    cx.append([t[0] for t in trip]); cy.append([t[1] for t in trip]); ck.append([t[2] for t in trip])
                lx.append(cx); ly.append(cy); lk.append(ck)
                a, b = h.root(l, c)
                rv.append(a); ri.append(b)
            Ix.append(lx); Iy.append(ly); Ik.append(lk); rootv.append(rv); rooti.append(ri)
        return Ix, Iy, Ik, rootv, rooti

    def argmin(self, rootv=None, rooti=None, Ix=None, Iy=None, Ik=None, capacity=4096) -> List[Candidate]:
        """argmin(parts, rootv, rooti, scales, Ix, Iy, Ik, candidates).  With no arguments it walks the tables min() left
        on the device; tables passed in (another engine's, or edited ones) are uploaded first and honoured."""
        h, g, m = self._h, self._h._geo, self._h.model
        if rootv is not None:
            for l in range(g["nlevels"]):
                for c in range(m.ncomponents):
                    h.set_root(l, c, rootv[l][c], rooti[l][c])
        if Ix is not None:
            for l in range(g["nlevels"]):
                for c in range(m.ncomponents):
                    for p in range(1, m.nparts(c)):
                        for pm in range(len(Ix[l][c][p])):
                            h.set_dp_pointers(l, c, p, pm, Ix[l][c][p][pm], Iy[l][c][p][pm], Ik[l][c][p][pm])
        return Candidate._unpack(*h.dp_argmin(capacity))


class PartsBasedDetector:
    """PartsBasedDetector<T> (include/PartsBasedDetector.hpp:152-175); dtype = np.float32 (src/demo.cpp:85)
    or np.float64 (ros/Node.hpp:121, cells/detect.cpp:93) picks the instantiation."""

    def __init__(self, device: int = 0, conv_mode: int = capi.PBD_CONV_AUTO, max_candidates: int = 4096,
                 level_begin: int = 0, level_end: int = 0, dtype=np.float32):
        self._device, self._conv, self._cap = device, conv_mode, max_candidates
        self._dtype = np.dtype(dtype)
        self._lb, self._le = level_begin, level_end
        self._h: Optional[capi.Handle] = None
        self._name = ""
        self.features_ = self.convolution_engine_ = self.dp_ = None

    def name(self) -> str:
        return self._name

    def distributeModel(self, model: Model) -> None:
        """src/PartsBasedDetector.cpp:102-127."""
        self._name = model.name
        self._h = capi.Handle(model, self._device, self._conv, self._cap, 0, self._lb, self._le, dtype=self._dtype)
        self.features_ = HOGFeatures(self._h)
        self.convolution_engine_ = SpatialConvolutionEngine(self._h)
        self.convolution_engine_.setFilters(model.filtersw)
        self.dp_ = DynamicProgram(self._h)

    @property
    def handle(self) -> capi.Handle:
        if self._h is None:
            raise RuntimeError("detect() before distributeModel()")
        return self._h

    def detect(self, im: np.ndarray, depth=None, candidates: Optional[List[Candidate]] = None) -> List[Candidate]:
        """src/PartsBasedDetector.cpp:69-95.  `depth` is accepted and ignored like the
        reference (:91-93); results are APPENDED to `candidates` (DynamicProgram.cpp:250)."""
        out = candidates if candidates is not None else []
        # (the image's dtype is its depth: uint8 -> pbd_detect_u8, the other accepted depths -> pbd_detect_image; unsupported ones raise)
        res = self.handle.detect(im, self._cap) if np.asarray(im).dtype == np.uint8 else self.handle.detect_image(im, self._cap)
        out.extend(Candidate._unpack(*res))
        return out
