#!/usr/bin/env python
"""Turn the rocprofv3 CSVs written by profiles/collect.sh into the committed summaries.

    python profiles/summarize.py gpurun_out/r01c r01c

writes profiles/<tag>_rocprof_summary.md, profiles/<tag>_kernel_stats*.csv, profiles/<tag>_bench_*.json
and refreshes profiles/traffic_dp.json (HBM bytes per frame of the dp_min stage from the PMC passes,
FETCH_SIZE x2 on gfx950 as MI355X_MICROARCH.md prescribes; per-pass frame counts are read from the trace).
"""
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
HERE = os.path.dirname(os.path.abspath(__file__))


def short(name):
    n = name.split("(")[0].replace("void ", "")
    return n


def stats_table(d):
    f = glob.glob(os.path.join(src, d, "*kernel_stats.csv")) + glob.glob(os.path.join(src, d, "*", "*kernel_stats.csv"))
    if not f:
        return None, None
    rows = list(csv.DictReader(open(f[0])))
    out = ["| kernel | calls | total ms | avg us | % | min us | max us |", "|---|---|---|---|---|---|---|"]
    for r in rows:
        out.append(f"| `{short(r['Name'])}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.2f} | "
                   f"{r['Percentage']} | {int(r['MinNs']) / 1e3:.2f} | {int(r['MaxNs']) / 1e3:.2f} |")
    return "\n".join(out), f[0]


def pmc(counter, prefix="pmc"):
    f = glob.glob(os.path.join(src, f"{prefix}_{counter}", "*counter_collection.csv")) + \
        glob.glob(os.path.join(src, f"{prefix}_{counter}", "*", "*counter_collection.csv"))
    if not f:
        return {}
    acc = {}
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        a = acc.setdefault(k, [0, 0.0, 0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return acc


def pmc_chains(counter, prefix):
    """dp_min launch chains of a pass in dispatch order (one frame in flight): each chain is the k_dt_pass launches up to and including
    a k_root launch; returns {k_root grid size: [chains, summed counter]} — the batch chains have the larger k_root grid."""
    f = glob.glob(os.path.join(src, f"{prefix}_{counter}", "*counter_collection.csv")) + \
        glob.glob(os.path.join(src, f"{prefix}_{counter}", "*", "*counter_collection.csv"))
    if not f:
        return {}
    rows = [r for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    out, cur = {}, 0.0
    for r in rows:
        k = short(r["Kernel_Name"])
        if k.startswith(("k_dt_pass", "k_reduce")):
            cur += float(r["Counter_Value"])
        elif k.startswith("k_root"):
            a = out.setdefault(int(r["Grid_Size"]), [0, 0.0])
            a[0] += 1
            a[1] += cur + float(r["Counter_Value"])
            cur = 0.0
    return out


md = [f"# {tag}: rocprofv3 summaries (MI355X, gfx950).  Collected by `profiles/collect.sh {tag}` (run from the repo root "
      "through gpurun), summarised by `profiles/summarize.py`.  bench.py = 26x6 person model, 640x480.", ""]
for d, title in (("stats_seq", "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --inflight 1 --no-cpu-baseline\n"
                  "(sequential frames: per-kernel durations undisturbed)"),
                 ("stats", "rocprofv3 --kernel-trace --stats -- python bench.py --graph 0 --no-prewarm --batch 1 --steps 20 --warmup 3 --no-cpu-baseline\n"
                  "(4 single frames in flight on 4 streams; kernels of different frames overlap)"),
                 ("stats_f64", "rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --inflight 1 --dtype f64 --no-cpu-baseline\n"
                  "(PartsBasedDetector<double>, sequential)"),
                 ("stats_batch", "rocprofv3 --kernel-trace --stats -- python bench.py --graph 0 --no-prewarm --steps 12 --warmup 3 --inflight 1 --batch 4 --no-cpu-baseline\n"
                  "(batches of 4 frames, one launch per stage for the batch, batches one at a time)")):
    t, path = stats_table(d)
    if t:
        md += [f"## {title}", t, ""]
        shutil.copy(path, os.path.join(HERE, f"{tag}_kernel_{d}.csv"))
fetch, write, clk = pmc("FETCH_SIZE"), pmc("WRITE_SIZE"), pmc("GRBM_GUI_ACTIVE")
if fetch:
    def frames_of(acc):   # every pass runs its own number of frames: one k_root launch per frame
        for k in acc:
            if k.startswith("k_root"):
                return max(1, acc[k][0])
        return 1
    nframes, nframes_w = frames_of(fetch), frames_of(write)
    md += ["## PMC passes (one counter per run): rocprofv3 --pmc <C> --kernel-trace -- python bench.py --steps 4 --warmup 2 --inflight 1 --no-cpu-baseline",
           f"({nframes} frames per run.)  FETCH_SIZE / WRITE_SIZE are in KB; gfx950 note (MI355X_MICROARCH.md, HBM): FETCH_SIZE "
           "under-reports coalesced streaming reads by 2x — RAW values here, `traffic_dp.json` applies the x2.  "
           "GRBM_GUI_ACTIVE sums the 8 XCDs: clock = value / 8 / duration.",
           "| kernel | calls | FETCH KB/call | WRITE KB/call | FETCH MB/frame (raw) | WRITE MB/frame | clock GHz |", "|---|---|---|---|---|---|---|"]
    for k in sorted(fetch, key=lambda k: -fetch[k][1]):
        f_, w_ = fetch[k], write.get(k, [1, 0.0, 1])
        c_ = clk.get(k)
        ghz = f"{c_[1] / 8 / c_[2]:.2f}" if c_ and c_[2] else "-"
        md.append(f"| `{k}` | {f_[0]} | {f_[1] / f_[0]:.1f} | {w_[1] / max(w_[0], 1):.1f} | {f_[1] / nframes / 1e3:.1f} | "
                  f"{w_[1] / nframes_w / 1e3:.1f} | {ghz} |")
    dpk = [k for k in fetch if k.startswith(("k_dt_pass", "k_reduce", "k_root"))]
    fb = sum(fetch[k][1] for k in dpk) * 1e3 / nframes
    wb = sum(write.get(k, [0, 0.0])[1] for k in dpk) * 1e3 / nframes_w
    tj = {"round": tag, "source": f"profiles/{tag}_rocprof_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, --inflight 1)",
          "kernels": sorted(dpk), "fetch_bytes_raw": fb, "write_bytes": wb,
          "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md HBM section); WRITE_SIZE uncalibrated, taken as is",
          "hbm_bytes_per_frame_corrected": 2 * fb + wb}
    md += ["", f"dp_min stage ({', '.join(sorted(dpk))}): fetch {fb / 1e6:.1f} MB raw (x2 = {2 * fb / 1e6:.1f} MB) + write {wb / 1e6:.1f} MB "
           f"= {(2 * fb + wb) / 1e6:.1f} MB per frame (algorithmic B_dp: see bench line)."]
    bn = os.path.join(src, "batch_n.txt")
    BF = int(open(bn).read().strip()) if os.path.exists(bn) else 3
    pfx = "pmcb" if os.path.exists(bn) else f"pmcb{BF}"
    fetch_b, write_b = pmc("FETCH_SIZE", pfx), pmc("WRITE_SIZE", pfx)
    if fetch_b and write_b:
        # the run also holds single-frame legs: keep the chains whose k_root grid is the batch's (the largest)
        cf, cw = pmc_chains("FETCH_SIZE", pfx), pmc_chains("WRITE_SIZE", pfx)
        gb = max(cf)
        nl = cf[gb][0]
        dpb = [k for k in fetch_b if k.startswith(("k_dt_pass", "k_reduce", "k_root"))]
        fbb = cf[gb][1] * 1e3 / nl
        wbb = cw[gb][1] * 1e3 / cw[gb][0]
        tj["batch"] = {"frames_per_launch": BF, "kernels": sorted(dpb), "fetch_bytes_raw": fbb, "write_bytes": wbb,
                       "hbm_bytes_per_launch_corrected": 2 * fbb + wbb, "hbm_bytes_per_frame_corrected": (2 * fbb + wbb) / BF}
        md += ["", f"the same passes with `--batch {BF}` (one launch chain per batch of {BF} frames, {nl} chains per run): dp_min stage fetch "
               f"{fbb / 1e6:.1f} MB raw (x2 = {2 * fbb / 1e6:.1f} MB) + write {wbb / 1e6:.1f} MB = {(2 * fbb + wbb) / 1e6:.1f} MB per launch chain "
               f"= {(2 * fbb + wbb) / BF / 1e6:.1f} MB per frame.",
               "| kernel | calls | FETCH KB/call | WRITE KB/call |", "|---|---|---|---|"]
        gs = min(cf)
        md[-2:] = [f"(single-frame chains of the same run, for comparison: {(2 * cf[gs][1] / cf[gs][0] + cw[gs][1] / cw[gs][0]) / 1e3:.1f} MB per frame.)"]
    json.dump(tj, open(os.path.join(HERE, "traffic_dp.json"), "w"), indent=1)
# SQ counters (one per pass) for the kernels of the dp_min stage and the filter bank
sq = {}
for d in sorted(glob.glob(os.path.join(src, "sq_*"))):
    c = os.path.basename(d)[3:]
    if not os.path.isdir(d):
        continue
    f = glob.glob(os.path.join(d, "*counter_collection.csv")) + glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    if not f:
        continue
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c:
            continue
        k = short(r["Kernel_Name"])
        if not k.startswith(("k_dt_pass", "k_reduce", "k_conv", "k_hog")):
            continue
        a = sq.setdefault(k, {}).setdefault(c, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
if sq:
    cs = sorted({c for k in sq for c in sq[k]})
    md += ["", "## SQ counters per launch (rocprofv3 --pmc <C> --kernel-trace, one counter per pass, sequential frames; SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* "
           "count quad-cycles summed over all waves)", "| kernel | " + " | ".join(cs) + " | ACTIVE_INST_ANY / WAVE_CYCLES | INSTS_VALU per WAVE_CYCLE |",
           "|---|" + "---|" * (len(cs) + 2)]
    for k in sorted(sq):
        v = {c: sq[k][c][1] / sq[k][c][0] for c in sq[k]}
        wc = v.get("SQ_WAVE_CYCLES", 0)
        md.append(f"| `{k}` | " + " | ".join(f"{v[c]:.3g}" if c in v else "-" for c in cs) +
                  f" | {v.get('SQ_ACTIVE_INST_ANY', 0) / wc if wc else 0:.3f} | {v.get('SQ_INSTS_VALU', 0) / wc if wc else 0:.3f} |")
cm = os.path.join(src, "conv_modes.json")
if os.path.exists(cm) and os.path.getsize(cm):
    j = json.load(open(cm))
    shutil.copy(cm, os.path.join(HERE, f"{tag}_conv_modes.json"))
    md += ["", f"## configs[4]: direct VALU correlation vs fp32 MFMA implicit GEMM ({j['workload']}; profiles/conv_modes.py, stage times from HIP events)",
           "| K | filters N | N x 800 | exact VALU ms | TFLOP/s | MFMA ms | TFLOP/s | PBD_CONV_AUTO |", "|---|---|---|---|---|---|---|---|"]
    for r in j["rows"]:
        md.append(f"| {r['mixtures']} | {r['filters']} | {r['contraction_NxK']} | {r['exact_valu_ms']} | {r['exact_valu_tflops']} | {r['mfma_f32_ms']} | "
                  f"{r['mfma_f32_tflops']} | {r['auto_picks']} |")
    t, path = stats_table("stats_conv_modes")
    if t:
        md += ["", "per-kernel view of the same script (rocprofv3 --kernel-trace --stats):", t]
        shutil.copy(path, os.path.join(HERE, f"{tag}_kernel_stats_conv_modes.csv"))
for t_ in ("sweep_sb.txt", "batch_stages.txt"):
    p = os.path.join(src, t_)
    if os.path.exists(p) and os.path.getsize(p):
        md += ["", f"## {t_}", "```", open(p).read().strip(), "```"]
for b in ("bench_n1.json", "bench_n1_driverflags.json", "bench_n1_b1.json", "bench_n1_inflight1.json", "bench_n1_f64.json"):
    p = os.path.join(src, b)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(HERE, f"{tag}_{b}"))
        md += ["", f"## {b}", "```", open(p).read().strip(), "```"]
open(os.path.join(HERE, f"{tag}_rocprof_summary.md"), "w").write("\n".join(md) + "\n")
print("wrote", os.path.join(HERE, f"{tag}_rocprof_summary.md"))
