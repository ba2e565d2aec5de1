"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol, argument /
state errors are reported through status codes (no GPU needed), candidate post-processing matches
the oracle, and the multi-GPU sharding + gather logic works over gloo with world_size 2."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from partsbaseddetector_amd import capi, parallel
from partsbaseddetector_amd.detector import Candidate
from partsbaseddetector_amd.model import make_image, make_person_model, make_tree_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "pbd_c.h")).read()
    declared = set(re.findall(r"\b(pbd_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("pbd_handle")
    assert len(declared) >= 30
    L = capi.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/pbd_c.h but not exported"
    assert declared == set(capi.EXPORTS)
    # the tuning build (same sources, planner knobs enabled; loaded by the tuning-variant parity test through PBD_LIBRARY)
    # must carry the same ABI: a stale copy fails here, not on the GPU box
    tune = os.path.join(os.path.dirname(capi.LIB_PATH), "libpbd_hip_tune.so")
    assert os.path.exists(tune), "libpbd_hip_tune.so missing: `make -C partsbaseddetector_amd/csrc` builds it next to libpbd_hip.so"
    Lt = C.CDLL(tune)
    for name in sorted(declared):
        assert hasattr(Lt, name), f"{name} missing from libpbd_hip_tune.so (stale build)"


def test_no_cpu_fallback_create_fails_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.PbdError) as e:
        capi.Handle(make_tree_model([-1, 0], 1))
    assert e.value.code == capi.PBD_ERR_HIP


def test_model_validation_errors():
    L = capi.lib()
    import torch
    for mutate, code in ((lambda m: m.defw.__setitem__((0, 0), 0.0), capi.PBD_ERR_ARG),      # a == 0: division by 2a
                         (lambda m: m.parentid[0].__setitem__(1, 2), capi.PBD_ERR_ARG),     # parent >= child
                         (lambda m: setattr(m, "flen", 31), capi.PBD_ERR_UNSUPPORTED)):
        m = make_tree_model([-1, 0, 1], 2, seed=1)
        if m.flen == 32 and code == capi.PBD_ERR_UNSUPPORTED:
            m.filtersw = [f[:, : 5 * 31].copy() for f in m.filtersw]
        mutate(m)
        with pytest.raises(capi.PbdError) as e:
            capi.Handle(m)
        assert e.value.code == code, e.value
    assert L.pbd_destroy(None) == capi.PBD_ERR_ARG
    assert L.pbd_detect_u8(None, None, 0, 0, 0, 0, None, None, None, 0, None) == capi.PBD_ERR_ARG


def test_candidates_sort_nms_match_oracle(orc):
    rng = np.random.default_rng(0)
    n, mp = 40, 5
    heads = np.zeros(n, capi.HEAD_DTYPE)
    heads["score"] = np.round(rng.normal(size=n), 1).astype(np.float32)   # ties -> stable order
    heads["nparts"] = mp
    boxes = np.zeros((n, mp, 4), np.int32)
    boxes[..., 0] = rng.integers(-10, 150, (n, mp)); boxes[..., 1] = rng.integers(-10, 110, (n, mp))
    boxes[..., 2:] = rng.integers(5, 40, (n, mp, 2))
    locs = rng.integers(0, 50, (n, mp, 3)).astype(np.int32)
    a = capi.candidates_sort(heads, boxes, locs)
    b = orc.candidates_sort(heads, boxes, locs)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    for ov in (0.0, 0.1, 0.5):
        ka = capi.candidates_nms(*a, 160, 120, ov)
        kb = orc.candidates_nms(*b, 160, 120, ov)
        assert len(ka[0]) == len(kb[0])
        for x, y in zip(ka, kb):
            np.testing.assert_array_equal(x, y)


def test_candidate_class_mirrors_reference():
    from partsbaseddetector_amd import Candidate
    c = [Candidate(np.array([[0, 0, 10, 10], [5, 5, 10, 10]]), np.array([s, 0], np.float32), 0, 0) for s in (0.1, 0.9, 0.5)]
    s = Candidate.sort(c)
    assert [round(x.score(), 1) for x in s] == [0.9, 0.5, 0.1]
    assert s[0].boundingBox() == (0, 0, 15, 15)
    kept = Candidate.nonMaximaSuppression((100, 100), s, 0.0)
    assert len(kept) == 1 and round(kept[0].score(), 1) == 0.9


def test_candidate_resize_and_setscore():
    """Candidate::resize / setScore (include/Candidate.hpp:76,82-89): integer fields times a float, truncated."""
    c = Candidate(np.array([[10, -7, 21, 5], [3, 4, 9, 9]], np.int32), np.array([1.5, 0.0], np.float32), 0)
    c.resize(0.5)
    np.testing.assert_array_equal(c.parts, [[5, -3, 10, 2], [1, 2, 4, 4]])   # -3.5 -> -3: toward zero
    c.setScore(2.25)
    assert c.score() == 2.25
    e = Candidate(np.zeros((0, 4), np.int32), np.zeros(0, np.float32), 0)
    e.setScore(-1.0)
    assert e.score() == -1.0


def test_level_and_frame_sharding(orc):
    g = orc.geometry(1920, 1080, 4, 10)
    cells = (g["cell_w"].astype(np.int64) * g["cell_h"]).tolist()
    for world in (1, 2, 4, 8):
        rng_ = parallel.shard_levels_contiguous(cells, world)
        assert rng_[0][0] == 0 and rng_[-1][1] == len(cells)
        assert all(rng_[i][1] == rng_[i + 1][0] for i in range(world - 1))
        loads = [sum(cells[b:e]) for b, e in rng_]
        assert max(loads) <= sum(cells) / world + max(cells)
    # LPT level sets (SURVEY 8e, configs[3]: one 1920x1080 frame on 8 GPUs; level 0 holds 13.1 % of the cells)
    for world in (1, 2, 4, 8):
        sets = parallel.shard_levels_lpt(cells, world)
        assert sorted(sum(sets, [])) == list(range(len(cells)))
        loads = [sum(cells[l] for l in s_) for s_ in sets]
        assert max(loads) <= max(max(cells), sum(cells) / world) * 1.05          # within 5 % of the lower bound
        assert max(loads) <= max(sum(cells[b:e]) for b, e in parallel.shard_levels_contiguous(cells, world))
    assert sum(cells) / max(sum(cells[l] for l in s_) for s_ in parallel.shard_levels_lpt(cells, 8)) > 7.3  # ideal speed-up
    assert parallel.shard_frames(32, 8, 3) == [3, 11, 19, 27]
    assert sorted(sum((parallel.shard_frames(32, 8, r) for r in range(8)), [])) == list(range(32))


def test_pack_unpack_roundtrip():
    rng = np.random.default_rng(1)
    n, mp = 17, 26
    heads = np.zeros(n, capi.HEAD_DTYPE)
    heads["score"] = rng.normal(size=n); heads["level"] = rng.integers(0, 46, n); heads["nparts"] = mp
    boxes = rng.integers(-100, 700, (n, mp, 4)).astype(np.int32)
    locs = rng.integers(0, 160, (n, mp, 3)).astype(np.int32)
    out = parallel.unpack_candidates(parallel.pack_candidates((heads, boxes, locs), mp, 64), mp)
    np.testing.assert_array_equal(out[0], heads); np.testing.assert_array_equal(out[1], boxes)
    np.testing.assert_array_equal(out[2], locs)


def test_dt_core_host(tmp_path):
    """partsbaseddetector_amd/csrc/dt_core.hpp — the segment-parallel distance transform k_dt_pass compiles — run on
    the host, lanes one after the other, against the oracle's sequential loop (tests/tools/dt_core_test.cpp): random,
    smooth, quantised (exact ties), sparse-peak, constant and concave lines, 1..64 lanes per line, lengths 1..700,
    float and double, every speculative-stitch order; every output and pointer bit-identical."""
    exe = tmp_path / "dt_core_test"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "partsbaseddetector_amd", "csrc"),
                           os.path.join(ROOT, "tests", "tools", "dt_core_test.cpp"), "-L", os.path.join(ROOT, "oracle"), "-lorc",
                           "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-o", str(exe)])
    out = subprocess.run([str(exe), "120000", "2026"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("bit-identical") == 2


def test_pack_candidates_overflow_raises():
    """A rank holding more candidates than the exchange buffer must fail loudly (the C ABI reports
    PBD_ERR_CAPACITY for the same condition), never drop detections in the multi-GPU merge."""
    heads = np.zeros(5, capi.HEAD_DTYPE)
    boxes, locs = np.zeros((5, 3, 4), np.int32), np.zeros((5, 3, 3), np.int32)
    with pytest.raises(OverflowError):
        parallel.pack_candidates((heads, boxes, locs), 3, 4)
    assert parallel.pack_candidates((heads, boxes, locs), 3, 5)[0] == 5


_WORKER = r"""
import os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, {root!r})
from partsbaseddetector_amd import capi, parallel
from partsbaseddetector_amd.detector import Candidate
from partsbaseddetector_amd.model import make_image, make_tree_model
from oracle import orc
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
model = make_tree_model([-1, 0, 0], 2, seed=6)
model.thresh = 1.8
frames = list(range(4))
mine = parallel.shard_frames(len(frames), world, rank)
# stand-in for the GPU detect of this rank's frames: the CPU oracle (this test covers the N>1 host path)
cands = [orc.detect(model, make_image(f, 120, 90))[:3] for f in mine]
merged_local = parallel.merge_candidates(cands)
gathered = parallel.gather_candidates(merged_local, 3, capacity=2048)
if rank == 0:
    allc = parallel.merge_candidates(gathered)
    ref = parallel.merge_candidates([orc.detect(model, make_image(f, 120, 90))[:3] for r in range(world) for f in parallel.shard_frames(4, world, r)])
    assert len(allc[0]) == len(ref[0]) > 0
    for a, b in zip(allc, ref):
        assert np.array_equal(a, b)
    print("GATHER_OK", len(allc[0]))
# gather to rank 0 only (what bench.py does every step inside the timed region)
g0 = parallel.gather_candidates(merged_local, 3, capacity=2048, dst=0)
assert (g0 is None) == (rank != 0)
if rank == 0:
    for a, b in zip(parallel.merge_candidates(g0), allc):
        assert np.array_equal(a, b)
    print("GATHER_DST_OK")
dist.barrier()
dist.destroy_process_group()
"""


def test_gloo_world2_frame_sharding_and_gather(tmp_path):
    """N>1 path on CPU: two ranks shard 4 frames, all_gather the candidate buffers (gloo)."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "GATHER_OK" in out.stdout and "GATHER_DST_OK" in out.stdout


def test_bench_respawns_itself_for_n_ranks():
    """`python bench.py --gpus N` outside torchrun launches its own N ranks on 127.0.0.1 with the same arguments."""
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.respawn_command(["--gpus", "8", "--steps", "20", "--warmup", "5"], 8, port=29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")


def test_traffic_file_matches_the_bench_defaults():
    """`roofline.traffic` of the default bench line comes from profiles/traffic_dp.json: its batch-chain entry has to be the
    one for bench.py's default batch size (else the line silently carries traffic = null), and its figures have to be
    self-consistent (FETCH_SIZE x2 + WRITE_SIZE; per-frame figures within 5 % of each other for single and batched chains)."""
    import json
    import re
    src = open(os.path.join(ROOT, "bench.py")).read()
    default_batch = int(re.search(r'os\.environ\.get\("PBD_BATCH", "(\d+)"\)', src).group(1))
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_dp.json")))
    assert tj["batch"]["frames_per_launch"] == default_batch
    assert abs(tj["hbm_bytes_per_frame_corrected"] - (2 * tj["fetch_bytes_raw"] + tj["write_bytes"])) < 1.0
    b = tj["batch"]
    assert abs(b["hbm_bytes_per_launch_corrected"] - (2 * b["fetch_bytes_raw"] + b["write_bytes"])) < 1.0
    assert abs(b["hbm_bytes_per_frame_corrected"] / tj["hbm_bytes_per_frame_corrected"] - 1.0) < 0.05


def test_batch_chain_trace_matches_the_bench_defaults():
    """The headline `roofline` must be recomputable from profiles/: the latest <tag>_chains_batch8.json (rocprofv3 kernel
    trace of a run launching only chains of the benched unit, profiles/collect_r04.sh) is for bench.py's default batch
    size, and the sum of its kernel durations agrees with the HIP-event `launch_ms` of the bench line of the same run."""
    import glob
    import json
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_chains_batch*.json")))
    if not files:
        pytest.skip("no batch-chain trace collected yet")
    src = open(os.path.join(ROOT, "bench.py")).read()
    default_batch = int(re.search(r'os\.environ\.get\("PBD_BATCH", "(\d+)"\)', src).group(1))
    ch = json.load(open(files[-1]))
    assert ch["frames_per_launch"] == default_batch
    g = ch["benched_chain"]
    assert g["chains"] >= 8 and g["launches_per_chain"] == 19          # 8 fold-x + 10 plain DT passes + k_root for the person tree
    ev = ch["bench_line_of_this_run"]["roofline.launch_ms (HIP events, same process, under the profiler)"]
    assert abs(g["sum_of_kernel_durations_ms"] / ev - 1.0) < 0.05, (g["sum_of_kernel_durations_ms"], ev)


def test_bench_line_schema_fields():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "schema", "batch_completion_ms", "child_legs"):
        assert f'"{key}"' in src
    assert '"frame_ms"' not in src        # schema 3: the per-batch completion pacing is no longer called a frame time


def test_ctypes_structs_match_the_c_header(tmp_path):
    """The ctypes mirrors of pbd_options / pbd_model_desc / pbd_candidate_head must have the C layout
    (compiled from include/pbd_c.h with gcc: sizes and the offsets of the fields added last)."""
    import ctypes as C
    import subprocess
    from partsbaseddetector_amd import model as M
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "pbd_c.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(pbd_options), offsetof(pbd_options, scalar_type),'
                   ' offsetof(pbd_options, reserved), sizeof(pbd_model_desc), offsetof(pbd_model_desc, biasid),'
                   ' sizeof(pbd_candidate_head)); return 0;}\n')
    exe = tmp_path / "layout"
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.check_call(["gcc", "-I", inc, "-o", str(exe), str(src)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    desc_t = type(M.make_tree_model([-1, 0], 1).to_desc())
    exp = [C.sizeof(capi.pbd_options), capi.pbd_options.scalar_type.offset, capi.pbd_options.reserved.offset,
           C.sizeof(desc_t), desc_t.biasid.offset, C.sizeof(capi.pbd_candidate_head)]
    assert got == exp, (got, exp)
    assert capi.PBD_SCALAR_F32 == 0 and capi.PBD_SCALAR_F64 == 1


def test_kernel_side_exact_divisions_by_multiplication():
    """The kernels replace integer divisions by wave-uniform divisors with multiply-high / multiply-shift constants
    (vector issue is what they have least of); the identities they rely on, over the ranges the sources state:
      * k_conv.hip: c / vw == (c * (65535 / vw + 1)) >> 16  for c < 256, 1 <= vw <= 16 (tile-local cell -> row);
      * pbd_internal.hpp dt_magic / k_dp.hip: x / d == umulhi(x, 0xFFFFFFFF / d + 1) for x * d < 2^32, d > 1 (lane -> line,
        element -> line of a block, segment starts)."""
    for vw in range(1, 17):
        m = 65535 // vw + 1
        assert all(((c * m) >> 16) == c // vw for c in range(256)), vw
    rng = np.random.default_rng(7)
    for d in list(range(2, 300)) + [511, 4096, 32767]:
        m = 0xFFFFFFFF // d + 1
        hi = (1 << 32) // d                      # x * d < 2^32
        xs = np.unique(np.concatenate([np.arange(0, min(hi, 2000)), rng.integers(0, hi, 2000), np.array([hi - 1, max(hi - d, 0)])]))
        for x in xs.tolist():
            assert (x * m) >> 32 == x // d, (x, d)

