#!/usr/bin/env python
"""bench.py — detect() frames/s, 640x480, 26-part x 6-mixture person model (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one PartsBasedDetector::detect() pass per GPU over one 640x480 synthetic frame that is
already resident in HBM (full pyramid: 46 levels; HOG -> filter bank -> DP min -> argmin; the
candidates are copied back to the host every step).  Frames are independent, so ranks shard frames
with no data-path collective ("weak" scaling: one frame per rank per step); the only collective is
the RCCL all_gather of the fixed-capacity candidate buffers after the timed loop has produced them
(SURVEY §8e).  Prints ONE JSON line on rank 0.
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--conv", choices=["auto", "exact", "mfma"], default="auto")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("PBD_INFLIGHT", "4")),
                    help="frames in flight per GPU on independent handles/streams (default 4: best measured "
                         "throughput with 8 hardware queues); 1 = strictly sequential detect() calls (latency mode)")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--mixtures", type=int, default=6)
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32",
                    help="instantiation: f32 = PartsBasedDetector<float> (BASELINE.json metric), f64 = <double> "
                         "(SURVEY 8f-3: the ROS node / ecto cell instantiation; exact VALU filter bank)")
    ap.add_argument("--shard", choices=["frames", "levels"], default="frames",
                    help="N>1: 'frames' = every rank its own frames (weak scaling, the BASELINE metric); 'levels' = all "
                         "ranks work on the SAME frames, each on an LPT-balanced set of pyramid levels (strong scaling, "
                         "BASELINE configs[3]: use with --width 1920 --height 1080)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo for CPU-side smoke runs)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from partsbaseddetector_amd import capi
    from partsbaseddetector_amd.model import make_image, make_person_model
    from partsbaseddetector_amd.parallel import gather_candidates

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
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
    conv = {"auto": capi.PBD_CONV_AUTO, "exact": capi.PBD_CONV_EXACT, "mfma": capi.PBD_CONV_MFMA}[args.conv]
    dtype = np.float64 if args.dtype == "f64" else np.float32
    # distinct frames per rank and per step slot (32 seeds as in configs[2]); resident in HBM
    nimg = 8
    by_levels = args.shard == "levels" and world > 1
    frames = [torch.from_numpy(make_image((0 if by_levels else rank * nimg) + i, W, H)).to(dev) for i in range(nimg)]
    torch.cuda.synchronize()
    model.thresh = pick_threshold(capi, model, torch.from_numpy(make_image(0, W, H)).to(dev), W, H, dtype=dtype)

    S = max(1, args.inflight)
    cap = 4096 if W * H <= 640 * 480 else 32768      # the 99.9th-percentile threshold scales the count with the area
    handles = [capi.Handle(model, device=local, conv_mode=conv, max_candidates=cap, dtype=dtype) for _ in range(S)]
    for hd in handles:
        hd.set_profiling(True)
    if by_levels:   # SURVEY 8e / configs[3]: one frame, levels spread over the ranks by greedy LPT on the cell counts
        from partsbaseddetector_amd.parallel import shard_levels_lpt
        g = handles[0].geometry(W, H)
        my_levels = shard_levels_lpt((g["cell_w"].astype(np.int64) * g["cell_h"]).tolist(), world)[rank]
        for hd in handles:
            hd.set_levels(my_levels)

    def run(nsteps, collect_out=None):
        pending = []
        for i in range(nsteps):
            hd = handles[i % S]
            if len(pending) == S:
                out = pending.pop(0).collect(cap)
                if collect_out is not None:
                    collect_out.append(out)
            hd.enqueue_dev(frames[i % nimg].data_ptr(), W, H, 3)
            pending.append(hd)
        for hd in pending:
            out = hd.collect(cap)
            if collect_out is not None:
                collect_out.append(out)

    run(args.warmup)
    for hd in handles:
        hd.dp_timer(reset=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = []
    run(args.steps, outs)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64)
        if cdev is not None:
            tmax = tmax.to(cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        # the one collective of the path: candidates of the last frame of every rank -> rank 0
        gathered = gather_candidates(outs[-1], handles[0].max_parts, capacity=1024, device=cdev)
        ncand_all = sum(len(g[0]) for g in gathered)
    else:
        ncand_all = len(outs[-1][0])

    # per-launch durations for the roofline: frames overlap when inflight > 1, which stretches every
    # kernel's span, so a short SEQUENTIAL leg (one frame in flight on one handle) is timed after the
    # throughput loop, with the same HIP events on the handle's stream.
    hd = handles[0]
    hd.dp_timer(reset=True)
    nseq = 20
    stage_acc = {}
    for i in range(nseq):
        hd.detect_dev(frames[i % nimg].data_ptr(), W, H, 3, capacity=cap)
        for k, v in hd.stage_ms().items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v / nseq
    dp_ms_seq = hd.dp_timer()[0]

    if rank == 0:
        work = hd.work()
        stage = stage_acc
        dp_ms = dp_ms_seq
        ms_per_step = dt / args.steps * 1e3
        value = args.steps * (1 if by_levels else world) / dt
        # roofline of the stage the north_star prices: the DP/distance-transform pass (HBM-bound).
        # achieved = algorithmic bytes of one frame's pass (B_dp) / its GPU time measured
        # with HIP events on the handle's stream.
        dp_gbs = work["B_dp"] / (dp_ms * 1e-3) / 1e9
        pdf_tf = work["F_pdf"] / (stage["pdf"] * 1e-3) / 1e12 if stage["pdf"] > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_dp.json")
        if os.path.exists(tpath) and (W, H, args.mixtures, args.dtype) == (640, 480, 6, "f32"):
            traffic = json.load(open(tpath))["hbm_bytes_per_frame_corrected"]
        if stage["dp_min"] >= stage["pdf"]:
            roof = {"kernel": "dp_min stage = 18 x k_dt_pass + 9 x k_reduce + k_root per frame", "bound": "hbm",
                    "achieved": round(dp_gbs, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(dp_gbs / 8000.0, 5),
                    "traffic": traffic, "launch_ms": round(float(dp_ms), 4), "algorithmic_bytes": work["B_dp"],
                    "timing": f"HIP events around the stage, mean of {nseq} sequential frames after the timed loop"}
        else:
            peak = 78.6 if args.dtype == "f64" else 157.3     # dense vector/matrix FMA peak of the dtype (MI355X_MICROARCH.md)
            roof = {"kernel": f"pdf filter bank (k_conv_mfma{'_f64, fp64' if args.dtype == 'f64' else ', fp32'} MFMA)" if conv != capi.PBD_CONV_EXACT
                    else f"pdf filter bank (k_conv_exact<{args.dtype}>, VALU, reference summation order)", "bound": "mfma",
                    "achieved": round(pdf_tf, 3),
                    "peak": peak, "unit": "TFLOP/s", "frac": round(pdf_tf / peak, 5), "traffic": None,
                    "launch_ms": round(float(stage["pdf"]), 4), "algorithmic_flops": work["F_pdf"]}
        line = {
            "metric": f"detect() frames/sec, {W}x{H}, 26-part person model",
            "value": round(value, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if by_levels else "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"person 26 parts x {args.mixtures} mixtures ({len(model.filtersw)} 5x5x32 filters), "
                                   f"{W}x{H} BGR, full pyramid ({hd.geometry(W, H)['nlevels']} levels), "
                                   f"threshold = 99.9th pct of root scores",
                       "frames_per_step_per_gpu": 1, "inflight": S, "conv": args.conv,
                       "candidates_last_frame": int(ncand_all), "parallelism": (f"levels (LPT sets) x{world}" if by_levels else f"frames x{world}")},
            "roofline": roof,
            "roofline_dt": {"bound": "hbm", "achieved": round(dp_gbs, 2), "peak": 8000.0, "unit": "GB/s",
                            "frac": round(dp_gbs / 8000.0, 5), "ms": round(float(dp_ms), 4)},
            "pdf": {"TFLOP/s": round(pdf_tf, 3), "peak": 78.6 if args.dtype == "f64" else 157.3,
                    "frac": round(pdf_tf / (78.6 if args.dtype == "f64" else 157.3), 4), "ms": round(stage["pdf"], 4)},
            "stage_ms_sequential": {k: round(v, 4) for k, v in stage.items()},
        }
        if not args.no_cpu_baseline:
            # bounded CPU sample: the oracle (reference-structured OpenMP restatement) on ONE frame
            from oracle import orc
            im = make_image(0, W, H)
            t = time.perf_counter()
            _, _, _, ms = orc.detect(model, im, dtype=dtype)[:4]
            cdt = time.perf_counter() - t
            line["cpu_baseline"] = {"value": round(1.0 / cdt, 4), "unit": "frames/s", "cores": orc.num_threads(),
                                    "kind": "port", "sample": "1 frame 640x480, same model, oracle/pbd_oracle.c "
                                    "(OpenMP at the reference's five loops)",
                                    "stage_ms": [round(x, 1) for x in ms]}
        print(json.dumps(line), flush=True)
    for hd in handles:
        hd.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
