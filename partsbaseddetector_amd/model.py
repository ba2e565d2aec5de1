"""Model container + synthetic model / image generators.

`Model` mirrors the reference's `Model` (include/Model.hpp:49-122): the same
fields with the same meaning, numpy instead of cv::Mat.  The reference's model
files are an un-vendored submodule (.gitmodules:1-3), so benchmarks and tests
use seeded synthetic models with the structure the reference's MATLAB side
produces (matlab/learning/buildmodel.m:19-80): see SURVEY.md §8(d).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List

import numpy as np


class pbd_model_desc(C.Structure):
    """ctypes mirror of include/pbd_c.h `pbd_model_desc`."""

    _fields_ = [
        ("nfilters", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32), ("flen", C.c_int32),
        ("norient", C.c_int32), ("sbin", C.c_int32), ("interval", C.c_int32), ("thresh", C.c_float),
        ("filters", C.POINTER(C.c_float)),
        ("ndefs", C.c_int32), ("defw", C.POINTER(C.c_float)), ("anchors", C.POINTER(C.c_int32)),
        ("nbias", C.c_int32), ("biasw", C.POINTER(C.c_float)),
        ("ncomponents", C.c_int32),
        ("part_offset", C.POINTER(C.c_int32)), ("parentid", C.POINTER(C.c_int32)),
        ("mix_offset", C.POINTER(C.c_int32)), ("filterid", C.POINTER(C.c_int32)),
        ("defid", C.POINTER(C.c_int32)), ("biasid", C.POINTER(C.c_int32)),
    ]


@dataclass
class Model:
    """include/Model.hpp:49-122.  Index vectors are 0-based (after zeroIndex)."""

    filtersw: List[np.ndarray]            # filters(): each kh x (kw*flen) float32, interleaved
    biasw: np.ndarray                     # bias()
    anchors: np.ndarray                   # anchors(): [ndefs, 2] (x, y)
    defw: np.ndarray                      # def(): [ndefs, 4]
    filterid: List[List[List[int]]]       # [component][part][mixture]
    biasid: List[List[List[int]]]
    defid: List[List[List[int]]]
    parentid: List[List[int]]             # [component][part]
    interval: int = 10                    # nscales()
    thresh: float = 0.0
    sbin: int = 4                         # binsize()
    norient: int = 18
    flen: int = 32
    name: str = "synthetic"
    _keep: list = field(default_factory=list, repr=False)

    @property
    def ncomponents(self) -> int:
        return len(self.filterid)

    def nparts(self, c: int) -> int:
        return len(self.filterid[c])

    @property
    def max_parts(self) -> int:
        return max(self.nparts(c) for c in range(self.ncomponents))

    def save(self, path: str) -> None:
        """Flat little-endian dump read by pbd::BinaryModel (partsbaseddetector_amd/host/pbd_host.hpp)."""
        import struct
        kh = self.filtersw[0].shape[0]
        kw = self.filtersw[0].shape[1] // self.flen
        with open(path, "wb") as f:
            f.write(b"PBDMODL1")
            f.write(struct.pack("<12i", len(self.filtersw), kh, kw, self.flen, self.norient, self.sbin, self.interval,
                                len(self.defw), len(self.biasw), self.ncomponents, 0, 0))
            f.write(struct.pack("<f", float(np.float32(self.thresh))))
            for w in self.filtersw:
                f.write(np.ascontiguousarray(w, np.float32).tobytes())
            f.write(np.ascontiguousarray(self.defw, np.float32).tobytes())
            f.write(np.ascontiguousarray(self.anchors, np.int32).tobytes())
            f.write(np.ascontiguousarray(self.biasw, np.float32).tobytes())
            for c in range(self.ncomponents):
                f.write(struct.pack("<i", self.nparts(c)))
                for p in range(self.nparts(c)):
                    k = len(self.filterid[c][p])
                    d = (list(self.defid[c][p]) + [0] * k)[:k] if p > 0 else [0] * k
                    b = list(self.biasid[c][p])
                    b = (b + [b[0] if b else 0] * k)[:k]
                    f.write(struct.pack("<2i", self.parentid[c][p] if p > 0 else -1, k))
                    f.write(np.asarray(self.filterid[c][p], np.int32).tobytes())
                    f.write(np.asarray(d, np.int32).tobytes())
                    f.write(np.asarray(b, np.int32).tobytes())

    @staticmethod
    def load(path: str) -> "Model":
        """Read the flat dump written by `save` / pbd::BinaryModel::serialize / pbd_modelconv."""
        import struct
        with open(path, "rb") as f:
            if f.read(8) != b"PBDMODL1":
                raise ValueError("not a PBDMODL1 file")
            nf, kh, kw, flen, norient, sbin, interval, ndefs, nbias, ncomp, _, _ = struct.unpack("<12i", f.read(48))
            thresh = struct.unpack("<f", f.read(4))[0]
            filt = [np.frombuffer(f.read(kh * kw * flen * 4), np.float32).reshape(kh, kw * flen).copy() for _ in range(nf)]
            defw = np.frombuffer(f.read(ndefs * 16), np.float32).reshape(ndefs, 4).copy()
            anchors = np.frombuffer(f.read(ndefs * 8), np.int32).reshape(ndefs, 2).copy()
            biasw = np.frombuffer(f.read(nbias * 4), np.float32).copy()
            filterid, defid, biasid, parentid = [], [], [], []
            for _c in range(ncomp):
                npart = struct.unpack("<i", f.read(4))[0]
                fi, di, bi, pa = [], [], [], []
                for p in range(npart):
                    par, k = struct.unpack("<2i", f.read(8))
                    fi.append(np.frombuffer(f.read(4 * k), np.int32).tolist())
                    d = np.frombuffer(f.read(4 * k), np.int32).tolist()
                    bi.append(np.frombuffer(f.read(4 * k), np.int32).tolist())
                    di.append(d if p > 0 else [])
                    pa.append(par)
                filterid.append(fi); defid.append(di); biasid.append(bi); parentid.append(pa)
        return Model(filt, biasw, anchors, defw, filterid, biasid, defid, parentid, interval, thresh, sbin, norient,
                     flen, os.path.basename(path))

    def save_filestorage(self, path: str) -> None:
        """Write the reference's on-disk format (cv::FileStorage layout of
        FileStorageModel::serialize, src/FileStorageModel.cpp:42-94) as OpenCV 2.4 emits it:
        XML for .xml, YAML for .yaml/.yml.  Used to exercise pbd::FileStorageModel::deserialize."""
        num = lambda v: (repr(float(np.float32(v))).replace("e-0", "e-0") if float(v) != int(v) else f"{int(v)}.")
        xml = path.endswith(".xml")
        o = []
        if xml:
            o += ['<?xml version="1.0"?>', "<opencv_storage>", f"<name>{self.name}</name>", f"<interval>{self.interval}</interval>",
                  f"<thresh>{num(self.thresh)}</thresh>", f"<sbin>{self.sbin}</sbin>", f"<norient>{self.norient}</norient>",
                  f"<flen>{self.flen}</flen>", "<filtersw>"]
            for w in self.filtersw:
                data = " ".join(num(v) for v in w.ravel())
                o += ['  <_ type_id="opencv-matrix">', f"    <rows>{w.shape[0]}</rows>", f"    <cols>{w.shape[1]}</cols>",
                      "    <dt>d</dt>", "    <data>", "      " + data + "</data></_>"]
            o += ["</filtersw>", "<biasw>", "  " + " ".join(num(v) for v in self.biasw) + "</biasw>", "<anchors>"]
            for a in np.asarray(self.anchors).reshape(-1, 2):
                o += ["  <_>", f"    {int(a[0])} {int(a[1])}</_>"]
            o += ["</anchors>", "<defs>"]
            for d in np.asarray(self.defw).reshape(-1, 4):
                o += ["  <_>", "    " + " ".join(num(v) for v in d) + "</_>"]
            o += ["</defs>", "<indexers>"]
            for c in range(self.ncomponents):
                o.append(f"  <component-{c}>")
                for p in range(self.nparts(c)):
                    di = self.defid[c][p] if p > 0 else []
                    o += [f"    <part-{p}>", f"      <parentid>{self.parentid[c][p] if p > 0 else 0}</parentid>",
                          "      <filterid>" + " ".join(map(str, self.filterid[c][p])) + "</filterid>",
                          "      <biasid>" + " ".join(map(str, self.biasid[c][p])) + "</biasid>",
                          "      <defid>" + " ".join(map(str, di)) + f"</defid></part-{p}>"]
                o.append(f"  </component-{c}>")
            o += ["</indexers>", "</opencv_storage>"]
        else:
            def flow(vals, per=8, ind="       "):
                vals = list(vals)
                rows = [", ".join(vals[i:i + per]) for i in range(0, len(vals), per)]
                return "[ " + (",\n" + ind).join(rows) + " ]"
            o += ["%YAML:1.0", f"name: {self.name}", f"interval: {self.interval}", f"thresh: {num(self.thresh)}",
                  f"sbin: {self.sbin}", f"norient: {self.norient}", f"flen: {self.flen}", "filtersw:"]
            for w in self.filtersw:
                o += ["   - !!opencv-matrix", f"      rows: {w.shape[0]}", f"      cols: {w.shape[1]}", "      dt: d",
                      "      data: " + flow([num(v) for v in w.ravel()], 6, "          ")]
            o += ["biasw: " + flow([num(v) for v in self.biasw]), "anchors:"]
            o += [f"   - [ {int(a[0])}, {int(a[1])} ]" for a in np.asarray(self.anchors).reshape(-1, 2)]
            o += ["defs:"] + ["   - " + flow([num(v) for v in d]) for d in np.asarray(self.defw).reshape(-1, 4)]
            o += ["indexers:"]
            for c in range(self.ncomponents):
                o.append(f"   component-{c}:")
                for p in range(self.nparts(c)):
                    di = self.defid[c][p] if p > 0 else []
                    o += [f"      part-{p}:", f"         parentid: {self.parentid[c][p] if p > 0 else 0}",
                          "         filterid: [ " + ", ".join(map(str, self.filterid[c][p])) + " ]",
                          "         biasid: [ " + ", ".join(map(str, self.biasid[c][p])) + " ]",
                          "         defid: [ " + ", ".join(map(str, di)) + " ]"]
        with open(path, "w") as f:
            f.write("\n".join(o) + "\n")

    def to_desc(self) -> pbd_model_desc:
        """Flatten into the C ABI descriptor (arrays kept alive on self)."""
        kh = self.filtersw[0].shape[0]
        kw = self.filtersw[0].shape[1] // self.flen
        for f in self.filtersw:
            if f.shape != (kh, kw * self.flen):
                raise ValueError("all filters must have the same size")
        filt = np.ascontiguousarray(np.stack(self.filtersw).astype(np.float32))
        defw = np.ascontiguousarray(np.asarray(self.defw, np.float32).reshape(-1, 4))
        anchors = np.ascontiguousarray(np.asarray(self.anchors, np.int32).reshape(-1, 2))
        biasw = np.ascontiguousarray(np.asarray(self.biasw, np.float32))
        part_offset, parentid, mix_offset, fid, did, bid = [0], [], [0], [], [], []
        for c in range(self.ncomponents):
            for p in range(self.nparts(c)):
                parentid.append(self.parentid[c][p] if p > 0 else -1)
                k = len(self.filterid[c][p])
                fid += list(self.filterid[c][p])
                d = list(self.defid[c][p]) if p > 0 else []
                did += (d + [0] * k)[:k]
                b = list(self.biasid[c][p])
                bid += (b + [b[0] if b else 0] * k)[:k]
                mix_offset.append(mix_offset[-1] + k)
            part_offset.append(part_offset[-1] + self.nparts(c))
        arrs = [np.asarray(a, np.int32) for a in (part_offset, parentid, mix_offset, fid, did, bid)]
        self._keep = [filt, defw, anchors, biasw] + arrs
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        d = pbd_model_desc()
        d.nfilters, d.kh, d.kw, d.flen = len(self.filtersw), kh, kw, self.flen
        d.norient, d.sbin, d.interval, d.thresh = self.norient, self.sbin, self.interval, float(self.thresh)
        d.filters = fp(filt)
        d.ndefs, d.defw, d.anchors = defw.shape[0], fp(defw), ip(anchors)
        d.nbias, d.biasw = biasw.shape[0], fp(biasw)
        d.ncomponents = self.ncomponents
        d.part_offset, d.parentid, d.mix_offset = ip(arrs[0]), ip(arrs[1]), ip(arrs[2])
        d.filterid, d.defid, d.biasid = ip(arrs[3]), ip(arrs[4]), ip(arrs[5])
        return d


# PARSE-style 26-part skeleton, 0-based parents (any tree with parent < child is valid)
PERSON_TREE = [-1, 0, 1, 2, 3, 4, 5, 2, 7, 8, 9, 10, 11, 12, 1, 14, 15, 16, 17, 14, 19, 20, 21, 22, 23, 24]


def _filters(rng, n, kh, kw, flen):
    w = rng.normal(0.0, 0.05, size=(n, kh, kw, flen)).astype(np.float32)
    w[..., flen - 1] = rng.normal(-0.1, 0.02, size=(n, kh, kw)).astype(np.float32)
    return [np.ascontiguousarray(w[i].reshape(kh, kw * flen)) for i in range(n)]


def _defs(rng, n, max_anchor=4):
    d = np.stack([rng.uniform(0.005, 0.05, n), rng.uniform(-0.01, 0.01, n),
                  rng.uniform(0.005, 0.05, n), rng.uniform(-0.01, 0.01, n)], axis=1).astype(np.float32)
    a = rng.integers(-max_anchor, max_anchor + 1, size=(n, 2)).astype(np.int32)
    return d, a


def make_tree_model(parents, K, seed=1234, kh=5, kw=5, sbin=4, interval=10, thresh=0.0, name="tree") -> Model:
    """One component, `len(parents)` parts, K mixtures per part.

    Layout of matlab/learning/buildmodel.m: filter p*K+k; child deformation
    (p-1)*K+k; child bias matrix L x K with biasid(l,k) = base + k*L + l; a
    single scalar root bias (= 0).
    """
    rng = np.random.default_rng(seed)
    P = len(parents)
    flen = 32
    filt = _filters(rng, P * K, kh, kw, flen)
    defw, anchors = _defs(rng, (P - 1) * K)
    L = K
    biasw = np.concatenate([[0.0], rng.normal(0.0, 0.1, (P - 1) * K * L)]).astype(np.float32)
    filterid = [[[p * K + k for k in range(K)] for p in range(P)]]
    defid = [[[] if p == 0 else [(p - 1) * K + k for k in range(K)] for p in range(P)]]
    biasid = [[[0] if p == 0 else [1 + (p - 1) * K * L + k * L for k in range(K)] for p in range(P)]]
    return Model(filt, biasw, anchors, defw, filterid, biasid, defid, [list(parents)], interval, thresh, sbin,
                 18, flen, name)


def make_person_model(seed=1234, K=6, thresh=0.0, interval=10, sbin=4) -> Model:
    """26 parts x K mixtures: 156 filters / 150 deformations / 901 biases for K=6."""
    return make_tree_model(PERSON_TREE, K, seed=seed, thresh=thresh, interval=interval, sbin=sbin,
                           name=f"person26x{K}")


def make_face_like_model(seed=77, ncomp=13, nfilters=146, part_counts=(39, 68), thresh=0.0, interval=5,
                         sbin=4) -> Model:
    """13 single-mixture components sharing one filter pool (matlab/modelTransfer.m:188-229):
    root bias per component, a dummy zero bias shared by all children."""
    rng = np.random.default_rng(seed)
    flen = 32
    filt = _filters(rng, nfilters, 5, 5, flen)
    filterid, defid, biasid, parentid = [], [], [], []
    ndefs = 0
    biasw = [0.0]  # index 0: the children's dummy zero bias
    for c in range(ncomp):
        P = part_counts[c % len(part_counts)]
        ids = rng.permutation(nfilters)[:P]
        par = [-1] + [int(rng.integers(max(0, p - 3), p)) for p in range(1, P)]
        filterid.append([[int(ids[p])] for p in range(P)])
        defid.append([[] if p == 0 else [ndefs + p - 1] for p in range(P)])
        ndefs += P - 1
        biasw.append(float(rng.normal(0.0, 0.1)))
        biasid.append([[len(biasw) - 1] if p == 0 else [0] for p in range(P)])
        parentid.append(par)
    defw, anchors = _defs(rng, ndefs, max_anchor=2)
    return Model(filt, np.asarray(biasw, np.float32), anchors, defw, filterid, biasid, defid, parentid, interval,
                 thresh, sbin, 18, flen, "face-like")


def make_image(seed: int, w: int, h: int, cn: int = 3) -> np.ndarray:
    """uint8 BGR test image: oriented gratings + filled rectangles/ellipses + noise,
    so every one of the 18 orientation bins and a wide range of magnitudes occur."""
    rng = np.random.default_rng(1000 + seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.full((h, w, 3), 110.0, np.float32)
    for _ in range(6):
        th = rng.uniform(0, np.pi)
        f = rng.uniform(0.02, 0.25)
        amp = rng.uniform(8, 30, size=3)
        ph = rng.uniform(0, 2 * np.pi)
        g = np.sin((xx * np.cos(th) + yy * np.sin(th)) * f * 2 * np.pi + ph)
        img += g[..., None] * amp[None, None, :]
    for _ in range(40):
        x0, y0 = rng.integers(0, w), rng.integers(0, h)
        sw, sh = rng.integers(4, max(5, w // 4)), rng.integers(4, max(5, h // 4))
        col = rng.uniform(0, 255, size=3)
        if rng.random() < 0.5:
            m = (xx >= x0) & (xx < x0 + sw) & (yy >= y0) & (yy < y0 + sh)
        else:
            m = ((xx - x0) / sw) ** 2 + ((yy - y0) / sh) ** 2 < 1.0
        a = rng.uniform(0.4, 1.0)
        img[m] = (1 - a) * img[m] + a * col
    img += rng.normal(0, 8, size=img.shape)
    out = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    if cn == 1:
        out = np.ascontiguousarray(out[..., 1])
    return np.ascontiguousarray(out)


def make_wide_image(kind, seed: int, w: int, h: int, cn: int = 3) -> np.ndarray:
    """A test image of one of the other depths the reference accepts (src/HOGFeatures.cpp:136-146) whose values USE the depth's range (not
    an 8-bit image in a wider container): np.uint16 -> the full 16 bits; np.float32 -> [0, 1] floats; np.float64 -> doubles incl. negative ones."""
    rng = np.random.default_rng(seed)
    base = make_image(seed, w, h, cn)
    kind = np.dtype(kind)
    if kind == np.uint16:
        return (base.astype(np.uint16) * 257) ^ rng.integers(0, 256, base.shape, dtype=np.uint16)
    noise = rng.uniform(-0.5, 0.5, base.shape)
    if kind == np.float32:
        return ((base + noise) / 255.0).astype(np.float32)
    if kind == np.float64:
        return (base + noise).astype(np.float64) * 3.0 - 100.0
    raise ValueError("make_wide_image: uint16, float32 or float64")
