#!/usr/bin/env python
"""bench.py — detect() frames/s, 640x480, 26-part x 6-mixture person model (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path per GPU over one BATCH of B synthetic 640x480 frames (default B = 16 —
round 5's last sessions: 3 x 16 measures 2.4-4.1 % above 3 x 8 on two boxes, 3 x 24 / 3 x 32 / 4 x 12 / 4 x 16 below it —:
pbd_detect_batch_enqueue_dev_u8 — every frame the full pyramid of 46 levels, HOG -> filter bank -> DP min -> argmin, one
launch per stage for the whole batch; the candidates of every frame are copied back to the host every step).  S steps
are in flight per GPU on S handles / streams (default S = 3).  --batch 1 = one frame per step (pbd_detect_enqueue_dev_u8).

Protocol:
  * setup (untimed, independent of --warmup): every handle and the clocks are pre-warmed for a fixed wall time;
  * W warm-up steps, then EXACTLY K timed steps between barrier + synchronize on both sides, max over ranks;
  * `value` = frames/s (K x B x ranks frames / that time) with the input frames ALREADY RESIDENT IN HBM when the timed
    region starts (the contract of this tier); the same K steps are then repeated handing over PINNED HOST images, so
    that every step contains its H2D copy (pbd_detect_batch_enqueue_u8): `value_incl_h2d`; and once more feeding the
    same handles one frame per call: `value_single_frame_calls`;
  * per-step wall time of the timed loop (completion-to-completion): median / p10 / p90;
  * a strictly sequential leg (one frame in flight, pbd_detect_u8-equivalent: H2D + kernels + D2H per call) gives
    the latency figures and the per-stage GPU times (HIP events on the handle's stream) behind
    `roofline_single_frame`; a second one runs BATCHES one at a time: `roofline` prices the dp_min launch chain of one
    batch (the unit the timed loop launches): B x B_dp algorithmic bytes / its HIP-event time; `traffic` = the PMC
    figure of the same chain from profiles/traffic_dp.json;
  * `cpu_baseline`: the oracle (reference-structured OpenMP restatement) on the box's host cores — 2 warm-ups +
    median of 5 frames on all cores, and a one-thread leg on a bounded sample.
Frames are independent, so ranks shard frames with no data-path collective ("weak" scaling: one batch per rank
per step); the only collective is the gather of the candidate buffers to rank 0 (RCCL over xGMI with backend
nccl) — done for EVERY step, inside the timed region ("multi-GPU wall time includes the RCCL
candidate gather").  `python bench.py --gpus N` without a torchrun environment spawns its own N ranks
(torch.distributed.run on 127.0.0.1); `--group` is the one-process alternative (pbd_group).
N > 1: only the timed legs run on every rank; the single-frame / sequential / batch-stage / CPU legs are rank 0's (the
other ranks wait at one barrier), and the line carries `config.ranks` (rank -> device index, PCI bus id, uuid, pid) and
`config.backend_world` gathered inside the run, so that a multi-GPU line can be audited from its JSON.
`--legs` restricts the run to some legs (profiling: `--legs batchseq --graph 0 --inflight 1` launches only batch chains,
one at a time — what `roofline` is quoted on; profiles/collect.sh).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Frames in flight run on one HIP stream each; the runtime multiplexes streams onto 4 hardware queues by
# default, which serialises the 4th stream behind another one.  Measured on MI355X: 4 frames in flight,
# 4 queues 727 frames/s, 8 queues 865 (3 in flight: 825 / 832).  Must be set before the HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

PREWARM_S = 1.5       # fixed wall time every rank spends running frames before --warmup / the timed region
# Version of the JSON line's keys.  1 = rounds 1-4 (a step = one batch of 8 on one handle).  2 = round 5 (a step = one batch on EACH of the S
# handles in flight, default batch 16: `frames_per_step_per_gpu`, `ms_per_step` changed meaning; `frame_ms` = per-batch completion pacing).
# 3 = round 6: `frame_ms` renamed `batch_completion_ms` (it never was a frame latency), `child_legs` reports how the child-process legs ended.
BENCH_SCHEMA = 3


def pick_threshold(capi, model, d_img, w, h, q=99.9, dtype=np.float32):
    """99.9-th percentile of the root scores of the seed frame, computed with the
    product path itself (pbd_get_root), so roughly 140 candidates per frame are back-tracked."""
    model.thresh = 3.0e38
    hd = capi.Handle(model, device=d_img.device.index, conv_mode=capi.PBD_CONV_AUTO, dtype=dtype)
    hd.detect_dev(d_img.data_ptr(), w, h, 3)
    hd._geo = hd.geometry(w, h)
    vals = np.concatenate([hd.root(l, 0)[0].ravel() for l in range(hd._geo["nlevels"])])
    hd.close()
    return float(np.float32(np.percentile(vals, q)))


def pct(a, q):
    return round(float(np.percentile(np.asarray(a, np.float64), q)), 4)


def cpu_info():
    """(model string, physical cores, hardware threads) of the host."""
    model, cores, phys, core = "unknown", set(), None, None
    try:
        for ln in open("/proc/cpuinfo"):
            k, _, v = ln.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None and core is not None:
                cores.add((phys, core)); phys = core = None
    except OSError:
        pass
    nthreads = os.cpu_count() or 1
    return model, (len(cores) or nthreads), nthreads


def respawn_command(argv, n, port=None):
    """`python bench.py --gpus N` outside torchrun: the command that runs the same arguments as N ranks on this node."""
    import socket
    if port is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def main_group(args):
    """bench.py --group --gpus N: the single-process path of SURVEY 8b / 8e (pbd_group_detect_batch_u8)."""
    import torch
    from partsbaseddetector_amd import capi
    from partsbaseddetector_amd.model import make_image, make_person_model
    W, H, N, S = args.width, args.height, args.gpus, max(1, args.inflight)
    if torch.cuda.device_count() < N:
        raise SystemExit(f"--gpus {N} but only {torch.cuda.device_count()} devices are visible")
    model = make_person_model(K=args.mixtures)
    dtype = np.float64 if args.dtype == "f64" else np.float32
    model.thresh = pick_threshold(capi, model, torch.from_numpy(make_image(0, W, H)).cuda(), W, H, dtype=dtype)
    conv = {"auto": capi.PBD_CONV_AUTO, "exact": capi.PBD_CONV_EXACT, "mfma": capi.PBD_CONV_MFMA, "split": capi.PBD_CONV_SPLIT,
            "split16": capi.PBD_CONV_SPLIT_F16}[args.conv]
    # every device listed S times (S frames in flight per device: host gather — RCCL wants distinct devices); S = 1: the RCCL all-gather
    # when librccl loads and N > 1 (PBD_GATHER_AUTO), and the line says which one ran and on how many ranks
    g = capi.Group(model, [d for _ in range(S) for d in range(N)], gather=capi.PBD_GATHER_HOST if S > 1 else capi.PBD_GATHER_AUTO, conv_mode=conv, dtype=dtype, graph=args.graph)
    pinned = [torch.from_numpy(make_image(i, W, H)).pin_memory() for i in range(8)]
    frames = [t.numpy() for t in pinned]
    t0 = time.perf_counter()
    nwarm = 0
    while time.perf_counter() - t0 < PREWARM_S and not args.no_prewarm:
        g.detect_batch([frames[i % 8] for i in range(4 * N * S)]); nwarm += 4 * N * S
    g.detect_batch([frames[i % 8] for i in range(max(1, args.warmup) * N)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = g.detect_batch([frames[i % 8] for i in range(args.steps * N)])
    dt = time.perf_counter() - t0
    line = {"metric": f"detect() frames/sec, {W}x{H}, 26-part person model", "value": round(args.steps * N / dt, 3), "unit": "frames/s",
            "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"person 26 parts x {args.mixtures} mixtures, {W}x{H} BGR, full pyramid, threshold = 99.9th pct of root scores",
                       "frames_per_step_per_gpu": 1, "inflight": S, "input": "pinned host images (H2D inside the timed region)",
                       "parallelism": f"pbd_group: one process, {N} device(s) x {S} members", "prewarm_frames": nwarm,
                       "group_size": g.size, "gather_mode": {capi.PBD_GATHER_HOST: "host", capi.PBD_GATHER_RCCL: "rccl"}.get(g.gather_mode, g.gather_mode),
                       "rccl_comm_size": g.comm_size, "frames_total": args.steps * N,
                       "devices": [{"device": d, "name": torch.cuda.get_device_properties(d).name,
                                    "pci_bus_id": getattr(torch.cuda.get_device_properties(d), "pci_bus_id", None),
                                    "uuid": str(getattr(torch.cuda.get_device_properties(d), "uuid", "")) or None} for d in range(N)],
                       "candidates_last_frame": int(len(outs[-1][0]))}}
    print(json.dumps(line), flush=True)
    g.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--conv", choices=["auto", "exact", "mfma", "split", "split16"], default="auto")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("PBD_INFLIGHT", "3")),
                    help="steps in flight per GPU on independent handles/streams (default 3 handles x batches of 16 frames: "
                         "best measured throughput); 1 = strictly sequential calls (latency mode)")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--mixtures", type=int, default=6)
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32",
                    help="instantiation: f32 = PartsBasedDetector<float> (BASELINE.json metric), f64 = <double> "
                         "(SURVEY 8f-3: the ROS node / ecto cell instantiation)")
    ap.add_argument("--shard", choices=["frames", "levels"], default="frames",
                    help="N>1: 'frames' = every rank its own frames (weak scaling, the BASELINE metric); 'levels' = all "
                         "ranks work on the SAME frames, each on an LPT-balanced set of pyramid levels (strong scaling, "
                         "BASELINE configs[3]: use with --width 1920 --height 1080)")
    ap.add_argument("--group", action="store_true",
                    help="ONE process driving --gpus N devices through pbd_group (no torchrun): every device listed --inflight "
                         "times, frames round-robin and software-pipelined over the members, host images in (H2D inside "
                         "the timed region), host gather of the candidates")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("PBD_BATCH", "16")),
                    help="frames per step and handle: >1 hands every handle a BATCH of same-sized frames (pbd_detect_batch_*: one "
                         "launch per stage for the whole batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("PBD_GRAPH", "1")), help="pbd_options.graph: replay a captured hipGraph per frame")
    ap.add_argument("--no-prewarm", action="store_true", help="skip the fixed pre-warm (profiling runs)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo for CPU-side smoke runs)")
    ap.add_argument("--legs", default="all",
                    help="comma-separated subset of timed,h2d,mfma32,split16,single,seq,batchseq,cpu (default all).  timed = the K steps `value` is quoted "
                         "on; h2d = the same with pinned host images; single = the handles fed one frame per call; seq = sequential "
                         "single frames with stage events; batchseq = batches one at a time with stage events (`roofline`); mfma32 = the "
                         "timed leg once more on handles with the fp32 MFMA filter bank (`value_fp32_mfma`, N = 1, when the timed handles run the split bank); cpu = the "
                         "oracle on the host cores.  Without `timed` the line's value is null (profiling runs)")
    args = ap.parse_args()
    ALL_LEGS = ("timed", "h2d", "mfma32", "split16", "single", "seq", "batchseq", "cpu")
    legs = set(ALL_LEGS) if args.legs == "all" else set(x for x in args.legs.split(",") if x)
    if legs - set(ALL_LEGS):
        raise SystemExit(f"--legs: unknown leg(s) {sorted(legs - set(ALL_LEGS))}; choose from {ALL_LEGS}")
    if args.no_cpu_baseline:
        legs.discard("cpu")

    import torch
    import torch.distributed as dist
    from partsbaseddetector_amd import capi
    from partsbaseddetector_amd.model import make_image, make_person_model
    from partsbaseddetector_amd.parallel import gather_candidates

    if args.group:
        return main_group(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no torchrun environment: become the launcher of our own N ranks (one process per GPU, rendezvous on 127.0.0.1)
        import subprocess
        if args.backend == "nccl" and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} device(s) visible "
                             f"(RCCL wants one GPU per rank; --backend gloo oversubscribes a GPU for smoke runs)")
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
        raise SystemExit(subprocess.call(respawn_command(sys.argv[1:], args.gpus), env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    local = local % max(1, torch.cuda.device_count())  # (gloo smoke runs may oversubscribe one GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    cdev = dev if args.backend == "nccl" else None      # where collective payloads live

    W, H = args.width, args.height
    model = make_person_model(K=args.mixtures)
    conv = {"auto": capi.PBD_CONV_AUTO, "exact": capi.PBD_CONV_EXACT, "mfma": capi.PBD_CONV_MFMA, "split": capi.PBD_CONV_SPLIT,
            "split16": capi.PBD_CONV_SPLIT_F16}[args.conv]
    dtype = np.float64 if args.dtype == "f64" else np.float32
    # distinct frames per rank and per step slot (32 seeds as in configs[2]); resident in HBM and, for the
    # H2D-inclusive leg, in pinned host memory
    B = max(1, args.batch)
    nimg = 32 if W * H <= 640 * 480 else 8           # (large frames: fewer, the buffers are 6 MB each)
    nimg = max(nimg // B, 1) * B if B <= nimg else B  # a whole number of distinct batches
    by_levels = args.shard == "levels" and world > 1
    host_frames = [torch.from_numpy(make_image((0 if by_levels else rank * nimg) + i, W, H)).pin_memory() for i in range(nimg)]
    frames = [t.to(dev) for t in host_frames]
    torch.cuda.synchronize()
    model.thresh = pick_threshold(capi, model, torch.from_numpy(make_image(0, W, H)).to(dev), W, H, dtype=dtype)

    S = max(1, args.inflight)
    cap = 4096 if W * H <= 640 * 480 else 32768      # the 99.9th-percentile threshold scales the count with the area
    handles = [capi.Handle(model, device=local, conv_mode=conv, max_candidates=cap * (B if B > 1 else 1), dtype=dtype, graph=args.graph) for _ in range(S)]
    nslots = nimg // B    # distinct step slots: slot k holds frames k * B .. k * B + B - 1 (no frame is in two slots)
    if B > 1:   # batches: B frames back to back in HBM (and B pinned host frames) per step slot
        dev_batches = [torch.stack([frames[k * B + j] for j in range(B)]).contiguous() for k in range(nslots)]
        host_batches = [[host_frames[k * B + j].data_ptr() for j in range(B)] for k in range(nslots)]
    if by_levels:   # SURVEY 8e / configs[3]: one frame, levels spread over the ranks by greedy LPT on the cell counts
        from partsbaseddetector_amd.parallel import shard_levels_lpt
        g = handles[0].geometry(W, H)
        level_cells = (g["cell_w"].astype(np.int64) * g["cell_h"]).tolist()
        level_sets = shard_levels_lpt(level_cells, world)
        my_levels = level_sets[rank]
        for hd in handles:
            hd.set_levels(my_levels)

    gathered_last = [None]
    B_gather = [B]      # frames per step of the leg that is running: a step's records are the candidates of ALL its frames (<= cap each)

    def gather(out):
        """The one collective of the path: this step's candidates of every rank -> rank 0 (N = 1: nothing to do)."""
        if world > 1:
            gathered_last[0] = gather_candidates(out, handles[0].max_parts, capacity=cap * B_gather[0], device=cdev, dst=0)

    def collect_one(hd, B):
        """the step's candidates: one frame's, or the batch's (concatenated: (level, component, root) order inside a frame)"""
        if B == 1:
            return hd.collect(cap)
        outs_b = hd.collect_batch(cap)
        return tuple(np.concatenate([o[k] for o in outs_b]) for k in range(3))

    def run(nsteps, collect_out=None, stamps=None, host=False, B=B, solo=False):
        """S frames in flight; host=True hands over pinned host images (H2D inside every step).  N > 1: every
        step's candidates are gathered to rank 0 — after the next frame has been enqueued, so the collective
        overlaps the GPU's work on the frames in flight.  solo: this rank alone (rank 0's extra legs): no collective."""
        pending = []
        for i in range(nsteps):
            hd = handles[i % S]
            out = None
            if len(pending) == S:
                out = collect_one(pending.pop(0), B)
                if stamps is not None:
                    stamps.append(time.perf_counter())
                if collect_out is not None:
                    collect_out.append(out)
            if B > 1:
                if host:
                    hd.enqueue_batch_host_ptrs(host_batches[i % nslots], W, H, 3)
                else:
                    hd.enqueue_batch_dev(dev_batches[i % nslots].data_ptr(), B, W, H, 3)
            elif host:
                hd.enqueue_host_ptr(host_frames[i % nimg].data_ptr(), W, H, 3)
            else:
                hd.enqueue_dev(frames[i % nimg].data_ptr(), W, H, 3)
            pending.append(hd)
            if out is not None and not solo:
                gather(out)
        for hd in pending:
            out = collect_one(hd, B)
            if stamps is not None:
                stamps.append(time.perf_counter())
            if collect_out is not None:
                collect_out.append(out)
            if not solo:
                gather(out)

    def timed(nsteps, host, B=B, solo=False):
        coll = world > 1 and not solo
        torch.cuda.synchronize()
        if coll:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs, stamps = [], [t0]
        run(nsteps * S, outs, stamps, host=host, B=B, solo=solo)      # a step = one batch on EACH of the S handles (see `config.step`)
        torch.cuda.synchronize()
        if coll:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if coll:
            tmax = torch.tensor([dt], dtype=torch.float64)
            if cdev is not None:
                tmax = tmax.to(cdev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, outs, np.diff(np.asarray(stamps)) * 1e3

    # ---- who is who (N > 1): gathered inside the run, so that the line proves which ranks ran on which devices ----
    ranks_info, backend_world = None, 1
    if world > 1:
        pr = torch.cuda.get_device_properties(local)
        me = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device": local, "name": pr.name,
              "pci_bus_id": getattr(pr, "pci_bus_id", None), "pci_device_id": getattr(pr, "pci_device_id", None),
              "pci_domain_id": getattr(pr, "pci_domain_id", None), "uuid": str(getattr(pr, "uuid", "")) or None,
              "pid": os.getpid(), "visible_devices": torch.cuda.device_count()}
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, me)
        backend_world = dist.get_world_size()

    # ---- setup: pre-warm every handle (plan, LDS opt-ins, pinned buffers) and the clocks for a fixed wall time ----
    prewarm_frames = 0
    if not args.no_prewarm:
        tw = time.perf_counter()
        while time.perf_counter() - tw < PREWARM_S or prewarm_frames < 4 * S:
            run(2 * S, host=(prewarm_frames // (2 * S)) % 2 == 1)
            prewarm_frames += 2 * S
    run(args.warmup * S)
    for hd in handles:
        hd.dp_timer(reset=True)

    # ---- the timed region: exactly K steps, frames resident in HBM ----
    dt = dt_h2d = None
    outs, per_frame_ms, per_frame_ms_h2d = [], [0.0], [0.0]
    if "timed" in legs:
        dt, outs, per_frame_ms = timed(args.steps, host=False)
    # ---- the same K steps handing over pinned host images: H2D inside every step ----
    if "h2d" in legs:
        dt_h2d, _, per_frame_ms_h2d = timed(args.steps, host=True)
    # ---- the timed leg once more with the fp32 MFMA filter bank (rounds 3-4's default) beside the split-product bank: same frames,
    #      same steps (bounded), fresh handles; N = 1 only ----
    CONV_NAMES = {capi.PBD_CONV_EXACT: "exact (VALU, reference summation order)", capi.PBD_CONV_MFMA: "mfma (fp32 / fp64 MFMA, k-ordered fma chain)",
                  capi.PBD_CONV_SPLIT: "split (fp32 products as six exact bfloat16 partial products on v_mfma_f32_32x32x16_bf16, fp32 accumulators)",
                  capi.PBD_CONV_SPLIT_F16: "split16 (OPT-IN: two scaled binary16 parts per operand, three products on v_mfma_f32_32x32x16_f16, fp32 accumulators; "
                                           "operands carried to 23 of 24 bits)"}
    conv_resolved = handles[0].conv_mode
    # (a SEPARATE process: fresh handles created beside the timed ones in this process stalled for tens of ms at a time — r05 session 2 —,
    #  and a second process repeats the protocol exactly: pre-warm, warm-up, K steps.  This process's handles are idle meanwhile.)
    value_mfma32, steps_mfma32 = None, min(args.steps, 100)
    child_status = {}     # leg -> how its child process ended (a failed child is reported, not a silent null)

    def run_child(leg, conv_name, legs_arg):
        """the same protocol in a second process with another filter bank; returns its JSON line or None, and records the exit"""
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(steps_mfma32), "--warmup", str(args.warmup), "--conv", conv_name,
               "--legs", legs_arg, "--inflight", str(S), "--batch", str(B), "--width", str(W), "--height", str(H), "--mixtures", str(args.mixtures),
               "--dtype", args.dtype, "--graph", str(args.graph)]
        torch.cuda.synchronize()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        except subprocess.TimeoutExpired:
            child_status[leg] = {"returncode": None, "error": "timeout after 300 s"}
            return None
        sub = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        child_status[leg] = {"returncode": r.returncode}
        if r.returncode != 0 or not sub:
            child_status[leg]["stderr_tail"] = r.stderr[-400:]
            return None
        try:
            return json.loads(sub[-1])
        except ValueError as e:
            child_status[leg]["error"] = f"unparsable line: {e}"
            return None

    if "mfma32" in legs and world == 1 and conv_resolved == capi.PBD_CONV_SPLIT:
        j = run_child("mfma32", "mfma", "timed")
        value_mfma32 = j.get("value") if j else None
    # ---- and with the opt-in binary16 bank (PBD_CONV_SPLIT_F16: half the matrix instructions; NOT the benched path) ----
    split16 = None
    if "split16" in legs and world == 1 and conv_resolved == capi.PBD_CONV_SPLIT:
        j = run_child("split16", "split16", "timed,batchseq")
        if j:
            split16 = {"value": j.get("value"), "unit": "frames/s", "steps": steps_mfma32, "pdf_ms_per_frame_batched": j.get("pdf", {}).get("ms_per_frame_batched"),
                       "pdf_fp32_equivalent_TFLOP/s": j.get("pdf", {}).get("TFLOP/s_batched"), "dp_min_ms_per_batch": j.get("roofline", {}).get("launch_ms"),
                       "what": "`bench.py --conv split16 --legs timed,batchseq` as a child process: PBD_CONV_SPLIT_F16, the OPT-IN bank of two scaled binary16 "
                               "parts per operand and three products (include/pbd_c.h; errors against fp64 measured equal to the benched bank's, operands carried "
                               "to 23 of their 24 bits) — reported beside `value`, never as it"}
    if world > 1:
        ncand_all = sum(len(g[0]) for g in gathered_last[0]) if (rank == 0 and gathered_last[0]) else 0   # the last step's gather (inside the timed region)
    else:
        ncand_all = len(outs[-1][0]) if outs else 0

    # Everything below describes ONE GPU (single-frame calls, sequential latency, stage times, roofline, CPU baseline): rank 0
    # measures it, the other ranks wait at the barrier — an N-rank run is N x the timed legs, not N x the whole script.
    extra = rank == 0
    dt_single = None
    seq_ms, stage_acc, dp_ms_seq, stage_batch, nseq = [0.0], {}, 0.0, None, 30
    hd = handles[0]
    if extra:
        # ---- the same handles called one frame at a time (pbd_detect_enqueue_dev_u8 / collect: S single frames in flight) ----
        if B > 1 and "single" in legs:
            run(3 * S, B=1, solo=True)                      # re-plan for single frames (untimed)
            dt_single, _, _ = timed(args.steps * B, host=False, B=1, solo=True)
        # ---- latency of ONE detect(): one frame in flight on one handle, host image in, candidates out (pbd_detect_u8 semantics), the way a
        #      caller runs it: hipGraph replay (pbd_options.graph), no stage events (rounds 1-4 quoted the figure of the profiled pass below:
        #      eager launches with an event after every stage) ----
        seq_ms_graph = None
        if "seq" in legs:
            seq_ms_graph = []
            for i in range(nseq + 5):
                t1 = time.perf_counter()
                hd.enqueue_host_ptr(host_frames[i % nimg].data_ptr(), W, H, 3)
                hd.collect(cap)
                t2 = time.perf_counter()
                if i >= 5:
                    seq_ms_graph.append((t2 - t1) * 1e3)
        hd.set_profiling(True)
        hd.dp_timer(reset=True)
        # ---- the same with per-stage HIP events (eager launches): the stage times of a frame on its own ----
        if "seq" in legs:
            seq_ms = []
            for i in range(nseq + 3):
                t1 = time.perf_counter()
                hd.enqueue_host_ptr(host_frames[i % nimg].data_ptr(), W, H, 3)
                hd.collect(cap)
                t2 = time.perf_counter()
                if i < 3:
                    hd.dp_timer(reset=True)
                    continue
                seq_ms.append((t2 - t1) * 1e3)
                for k, v in hd.stage_ms().items():
                    stage_acc[k] = stage_acc.get(k, 0.0) + v / nseq
            dp_ms_seq = hd.dp_timer()[0]
        # ---- the benched unit of work, one at a time: a batch of B frames per call, per-stage HIP events on (eager launches) ----
        if B > 1 and "batchseq" in legs:
            nbat = 12
            stage_batch = {}
            for i in range(nbat + 3):
                hd.enqueue_batch_dev(dev_batches[i % nslots].data_ptr(), B, W, H, 3)
                hd.collect_batch(cap)
                if i >= 3:
                    for k, v in hd.stage_ms().items():
                        stage_batch[k] = stage_batch.get(k, 0.0) + v / nbat
        hd.set_profiling(False)

    if rank == 0:
        if not stage_acc and stage_batch is None:   # neither stage leg ran: one plan is still needed for work()
            hd.enqueue_dev(frames[0].data_ptr(), W, H, 3); hd.collect(cap)
        work = hd.work()
        stage = stage_acc or {k: 0.0 for k in ("image_pyramid", "hog", "pdf", "dp_min", "argmin", "total")}
        dp_ms = dp_ms_seq
        per_rank = (1 if by_levels else world) * B * S     # frames per step: every rank's S handles take one batch of B frames each
        ms_per_step = dt / args.steps * 1e3 if dt else None
        value = args.steps * per_rank / dt if dt else None
        value_h2d = args.steps * per_rank / dt_h2d if dt_h2d else None
        # roofline of the stage the north_star prices: the DP/distance-transform pass (HBM-bound by bytes).
        # achieved = algorithmic bytes of one frame's pass (B_dp) / its GPU time measured
        # with HIP events on the handle's stream in the sequential leg.
        dp_gbs = work["B_dp"] / (dp_ms * 1e-3) / 1e9 if dp_ms > 0 else 0.0
        pdf_tf = work["F_pdf"] / (stage["pdf"] * 1e-3) / 1e12 if stage["pdf"] > 0 else 0.0
        traffic, traffic_source, traffic_b = None, None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_dp.json")
        if os.path.exists(tpath) and (W, H, args.mixtures, args.dtype) == (640, 480, 6, "f32"):
            tj = json.load(open(tpath))
            traffic = tj["hbm_bytes_per_frame_corrected"]
            traffic_source = (f"profiles/traffic_dp.json ({tj.get('round', '?')}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                              f"command, FETCH x2 per MI355X_MICROARCH.md; committed file, not measured in this run)")
            tb = tj.get("batch")
            if tb and tb.get("frames_per_launch") == B:
                traffic_b = tb["hbm_bytes_per_launch_corrected"]
        roof_single = {"kernel": "dp_min stage (distance-transform passes + root), one frame per launch chain", "bound": "hbm",
                       "achieved": round(dp_gbs, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(dp_gbs / 8000.0, 5),
                       "traffic": traffic, "traffic_source": traffic_source, "launch_ms": round(float(dp_ms), 4),
                       "algorithmic_bytes": work["B_dp"], "launch_mode": "eager launches, per-stage HIP events on (the timed loop replays a hipGraph)",
                       "timing": f"HIP events around the stage, mean of {nseq} sequential frames after the timed loop"}
        if stage_batch is not None and stage_batch["dp_min"] >= stage_batch["pdf"]:
            # the launch chain of the benched workload covers B frames (virtual pyramid levels): bytes per launch = B x B_dp
            gbs_b = B * work["B_dp"] / (stage_batch["dp_min"] * 1e-3) / 1e9
            roof = {"kernel": f"dp_min stage (distance-transform passes + root; the children's mixture reduce folded into the parents' x pass), "
                              f"one launch chain per batch of {B} frames", "bound": "hbm",
                    "achieved": round(gbs_b, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs_b / 8000.0, 5),
                    "traffic": traffic_b, "traffic_source": traffic_source if traffic_b else None,
                    "launch_ms": round(float(stage_batch["dp_min"]), 4), "units_per_launch": B,
                    "algorithmic_bytes": B * work["B_dp"], "algorithmic_bytes_per_frame": work["B_dp"],
                    "launch_mode": "eager launches, per-stage HIP events on (the timed loop replays a hipGraph)",
                    "timing": "HIP events around the stage, mean of 12 sequential batches after the timed loop; the rocprofv3 kernel trace of the "
                              "same leg (bench.py --legs batchseq --graph 0 --inflight 1) is profiles/*_kernel_stats_batch<B>.csv"}
        elif stage["dp_min"] >= stage["pdf"]:
            roof = roof_single
        else:
            peak = 78.6 if args.dtype == "f64" else 157.3     # dense vector/matrix FMA peak of the dtype (MI355X_MICROARCH.md); fp32-equivalent flops for the split bank
            roof = {"kernel": f"pdf filter bank (k_conv_mfma{'_f64, fp64' if args.dtype == 'f64' else ', fp32'} MFMA)" if conv != capi.PBD_CONV_EXACT
                    else f"pdf filter bank (k_conv_exact<{args.dtype}>, VALU, reference summation order)", "bound": "mfma",
                    "achieved": round(pdf_tf, 3),
                    "peak": peak, "unit": "TFLOP/s", "frac": round(pdf_tf / peak, 5), "traffic": None,
                    "launch_ms": round(float(stage["pdf"]), 4), "algorithmic_flops": work["F_pdf"]}
        # filter bank: fp32-equivalent flops (F_pdf) per second; the split bank executes 6 bf16 MFMA flops per fp32-equivalent one and is
        # priced against the dense bf16 peak (2.5 PF), the fp32 / fp64 MFMA banks against theirs
        is_split = conv_resolved in (capi.PBD_CONV_SPLIT, capi.PBD_CONV_SPLIT_F16)
        pdf_peak = 2500.0 if is_split else (78.6 if args.dtype == "f64" else 157.3)       # (bf16 and f16 dense MFMA peaks are the same)
        pdf_mult = 6.0 if conv_resolved == capi.PBD_CONV_SPLIT else 3.0 if is_split else 1.0
        pdf_ms_b = stage_batch["pdf"] / B if stage_batch else None
        pdf_tf_b = work["F_pdf"] / (pdf_ms_b * 1e-3) / 1e12 if pdf_ms_b else None
        pdf_block = {"TFLOP/s": round(pdf_tf, 3), "what": "fp32-equivalent TFLOP/s (F_pdf = 2 x cells x filters x kh kw 32), a frame on its own",
                     "ms": round(stage["pdf"], 4), "ms_per_frame_batched": (round(pdf_ms_b, 4) if pdf_ms_b else None),
                     "TFLOP/s_batched": (round(pdf_tf_b, 3) if pdf_tf_b else None),
                     "mfma_TFLOP/s_batched": (round(pdf_tf_b * pdf_mult, 2) if pdf_tf_b else None), "mfma_flops_per_fp32_flop": pdf_mult,
                     "peak": pdf_peak, "frac": round((pdf_tf_b if pdf_tf_b else pdf_tf) * pdf_mult / pdf_peak, 4),
                     "bank": CONV_NAMES.get(conv_resolved, str(conv_resolved))}
        rnd = lambda v, n=3: None if v is None else round(v, n)
        config = {"workload": f"person 26 parts x {args.mixtures} mixtures ({len(model.filtersw)} {model.filtersw[0].shape[0]}x{model.filtersw[0].shape[1] // 32}x32 filters), "
                              f"{W}x{H} BGR, full pyramid ({hd.geometry(W, H)['nlevels']} levels), "
                              f"threshold = 99.9th pct of root scores",
                  "frames_per_step_per_gpu": B * S, "frames_per_batch": B, "inflight": S,
                  "step": (f"one batch of {B} frames on each of the {S} handles in flight = {B * S} frames per GPU (rounds 1-4 counted every batch as a step: with the "
                           f"driver's 20 steps the timed window was 0.08-0.10 s and the pipeline's fill and drain — one batch running alone at either end — weighed 4-5 %; "
                           f"a step of {S} batches times {S}x the frames between the same two synchronisations)"), "conv": CONV_NAMES.get(conv_resolved, str(conv_resolved)), "conv_requested": args.conv,
                  "frames_total_per_step": B * S * (1 if by_levels else world), "input": "frames resident in HBM",
                  "distinct_frames_per_rank": nimg, "distinct_step_slots": nslots,
                  "batching": (f"pbd_detect_batch: every handle processes {B} frames per step, one launch per stage for the batch" if B > 1 else "single frames"),
                  "launch": "hipGraph replay (one hipGraphLaunch per step)" if args.graph else "eager (~30 launches per step)",
                  "legs": sorted(legs), "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                  "prewarm_s": 0.0 if args.no_prewarm else PREWARM_S, "prewarm_frames": prewarm_frames,
                  "candidates_last_step": int(ncand_all), "candidates_last_step_is": "records of the last timed step, all its frames (and all ranks: the gathered list)", "parallelism": (f"levels (LPT sets) x{world}" if by_levels else f"frames x{world}"),
                  "gather": (f"every step, inside the timed region: torch.distributed gather to rank 0, backend {args.backend}, "
                             f"counts first, then the records padded to the longest list" if world > 1 else "none (one rank)")}
        if by_levels:
            # what the LPT partition is: every rank's levels and share of the frame's cells; the makespan (largest share) against
            # the ideal 1 / N — level 0 alone is a fixed fraction of the cells, which bounds the strong-scaling speed-up
            tot = float(sum(level_cells))
            shares = [sum(level_cells[l] for l in ls) / tot for ls in level_sets]
            config["level_sets"] = [{"rank": r, "levels": [int(l) for l in ls], "cell_share": round(sh, 4)} for r, (ls, sh) in enumerate(zip(level_sets, shares))]
            config["lpt_makespan_share"] = round(max(shares), 4)
            config["lpt_speedup_bound"] = round(1.0 / max(shares), 3)
            config["lpt_ideal_speedup"] = world
        if world > 1:
            config["backend"] = dist.get_backend()
            config["backend_world"] = backend_world
            config["ranks"] = ranks_info
            config["distinct_devices"] = len({(r["device"], r["pci_bus_id"], r["uuid"]) for r in ranks_info})
            config["rank0_only_legs"] = "single-frame calls, sequential latency, stage times / roofline (the other ranks wait at a barrier); the CPU baseline is measured by N = 1 runs only"
        line = {
            "metric": f"detect() frames/sec, {W}x{H}, 26-part person model",
            "value": rnd(value), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": rnd(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if by_levels else "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "schema": BENCH_SCHEMA,
            "config": config,
            "batch_completion_ms": {"median": pct(per_frame_ms, 50), "p10": pct(per_frame_ms, 10), "p90": pct(per_frame_ms, 90),
                         "what": f"completion-to-completion wall time per BATCH in the timed loop (rank 0): completions of the {S} batches in flight "
                                 f"arrive in bursts — throughput pacing, not latency (latency: `sequential`)"},
            "value_resident": rnd(value), "value_incl_h2d": rnd(value_h2d),
            "value_fp32_mfma": value_mfma32,
            "value_fp32_mfma_is": (f"`value` of `bench.py --conv mfma --legs timed --steps {steps_mfma32}` (PBD_CONV_MFMA: the fp32 v_mfma_f32_16x16x4_f32 bank, the default of "
                                   f"rounds 3-4), run as a child process after this process's timed legs" if value_mfma32 else None),
            "opt_in_split_f16": split16,
            "child_legs": child_status or None,
            "value_single_frame_calls": (round(args.steps * B * S / dt_single, 3) if dt_single else None),
            "value_is": "frames resident in HBM when the timed region starts (the tier's contract, DESIGN.md 7); value_incl_h2d = the same steps "
                        "from pinned host images; value_single_frame_calls = ONE GPU's handles fed one frame per call",
            "incl_h2d": {"value": rnd(value_h2d), "unit": "frames/s", "ms_per_step": rnd(dt_h2d / args.steps * 1e3 if dt_h2d else None, 4),
                         "batch_completion_ms": {"median": pct(per_frame_ms_h2d, 50), "p10": pct(per_frame_ms_h2d, 10), "p90": pct(per_frame_ms_h2d, 90)},
                         "what": "the same K steps with every frame handed over as a pinned host image (pbd_detect_[batch_]enqueue_u8: "
                                 "H2D + kernels + D2H of the candidates per step)"},
            "sequential": {"latency_ms": ({"median": pct(seq_ms_graph, 50), "p10": pct(seq_ms_graph, 10), "p90": pct(seq_ms_graph, 90)} if seq_ms_graph else None),
                           "latency_ms_profiled_pass": {"median": pct(seq_ms, 50), "p10": pct(seq_ms, 10), "p90": pct(seq_ms, 90)},
                           "frames": nseq, "what": "one frame in flight: host image in, candidates out, wall time per call; latency_ms = as a caller runs it "
                                                   "(graph replay when pbd_options.graph is set, no events), latency_ms_profiled_pass = the eager pass with per-stage "
                                                   "events that `stage_ms_sequential` comes from (the figure rounds 1-4 quoted)"},
            "roofline": roof,
            "roofline_single_frame": roof_single,
            "roofline_dt": {"bound": "hbm", "achieved": round(dp_gbs, 2), "peak": 8000.0, "unit": "GB/s",
                            "frac": round(dp_gbs / 8000.0, 5), "ms": round(float(dp_ms), 4)},
            "pdf": pdf_block,
            "stage_ms_sequential": {k: round(v, 4) for k, v in stage.items()},
            "stage_ms_per_frame_batched": ({k: round(v / B, 4) for k, v in stage_batch.items()} if stage_batch else None),
        }
        if "cpu" in legs and world == 1:       # (the CPU baseline is an N = 1 figure: an N-rank run does not spend 15 s on it)
            # bounded CPU sample: the oracle (reference-structured OpenMP restatement), same model and image size.
            from oracle import orc
            ims = [make_image(i, W, H) for i in range(3)]
            cpu_name, ncores, nhw = cpu_info()
            orc.set_num_threads(ncores)                # one thread per physical core (SMT siblings only slow it down)
            times, stage_ms = [], None
            for i in range(2 + 5):                     # 2 warm-ups + 5 timed frames, median
                t = time.perf_counter()
                _, _, _, ms = orc.detect(model, ims[i % 3], dtype=dtype)[:4]
                if i >= 2:
                    times.append(time.perf_counter() - t)
                    stage_ms = ms
            med = float(np.median(times))
            # one-thread leg (OMP_NUM_THREADS=1 equivalent) on a bounded sample: ONE frame, same model and size
            orc.set_num_threads(1)
            t = time.perf_counter()
            orc.detect(model, ims[0], dtype=dtype)
            t1 = time.perf_counter() - t
            orc.set_num_threads(ncores)
            flags = "unknown"
            try:
                for ln in open(os.path.join(ROOT, "oracle", "Makefile")):
                    if ln.startswith("CFLAGS"):
                        flags = ln.split("=", 1)[1].strip()
            except OSError:
                pass
            line["cpu_baseline"] = {"value": round(1.0 / med, 4), "unit": "frames/s", "cores": ncores, "threads": ncores,
                                    "hw_threads": nhw, "kind": "port", "cpu": cpu_name,
                                    "flags": f"gcc {flags} (built where the repository is built: no -march=native, the binary travels to the "
                                             f"GPU box); OpenMP: plain `#pragma omp parallel for` (static schedule) at the reference's five sites",
                                    "sample": f"median of 5 frames {W}x{H} after 2 warm-ups, same model, oracle/pbd_oracle.c "
                                              f"(the filter bank is `omp parallel for` over its {len(model.filtersw)} filters on "
                                              f"{ncores} threads, as src/SpatialConvolutionEngine.cpp:114-117), one thread on each of the {ncores} physical cores",
                                    "frame_s": {"median": round(med, 4), "min": round(min(times), 4), "max": round(max(times), 4)},
                                    "stage_ms": [round(x, 1) for x in stage_ms],
                                    "single_thread": {"value": round(1.0 / t1, 4), "unit": "frames/s", "threads": 1,
                                                      "sample": f"1 frame {W}x{H}, same model (OMP_NUM_THREADS=1 equivalent)"}}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()       # ranks > 0 have been waiting here while rank 0 ran its extra legs
    for hd in handles:
        hd.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
