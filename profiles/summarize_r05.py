#!/usr/bin/env python
"""Turn what profiles/collect_r05.sh wrote under gpurun_out/<tag>/ into the committed evidence:

    python profiles/summarize_r05.py gpurun_out/r04a r04a

  profiles/<tag>_kernel_stats_batch8.csv   per (kernel, grid): the run that launches only batch-of-8 chains (rocprofv3 --kernel-trace)
  profiles/<tag>_chains_batch8.json        the dp_min launch chains of that run: sum of kernel durations / span per chain,
                                           `frames_per_launch`, and the bench line's HIP-event `launch_ms` of the SAME run beside it
  profiles/<tag>_kernel_stats_seq.csv, _chains_seq.json   the same for single frames
  profiles/traffic_dp.json                 HBM bytes of the dp_min chain from the FETCH_SIZE / WRITE_SIZE passes (FETCH x2 on gfx950)
  profiles/<tag>_rocprof_summary.md        tables + the bench lines
"""
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
HERE = os.path.dirname(os.path.abspath(__file__))
BF = 8      # frames per launch chain of the batch runs: read from the trace run's own bench line below (config.frames_per_batch)


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def counter_rows(d):
    f = glob.glob(os.path.join(src, d, "*counter_collection.csv")) + glob.glob(os.path.join(src, d, "*", "*counter_collection.csv"))
    return list(csv.DictReader(open(f[0]))) if f else []


def chains_of(rows, counter):
    """{k_root grid: [chains, summed counter over the chain's k_dt_pass / k_root launches]} in dispatch order (one chain in flight)"""
    rows = sorted((r for r in rows if r["Counter_Name"] == counter), key=lambda r: int(r["Dispatch_Id"]))
    out, cur = {}, 0.0
    for r in rows:
        k = short(r["Kernel_Name"])
        if k.startswith(("k_dt_pass", "k_reduce")):
            cur += float(r["Counter_Value"])
        elif k.startswith("k_root"):
            a = out.setdefault(int(r["Grid_Size"]), [0, 0.0])
            a[0] += 1; a[1] += cur + float(r["Counter_Value"]); cur = 0.0
    return out


def last_json_line(path):
    if not os.path.exists(path):
        return None
    for ln in reversed(open(path).read().splitlines()):
        if ln.startswith("{"):
            return json.loads(ln)
    return None


_l = None
for _f in ("trace8.json", "bench_n1_driverflags.json", "bench_n1.json"):
    if os.path.exists(os.path.join(src, _f)):
        for _ln in reversed(open(os.path.join(src, _f)).read().splitlines()):
            if _ln.startswith("{"):
                _l = json.loads(_ln); break
    if _l:
        break
if _l and _l.get("config", {}).get("frames_per_batch"):
    BF = int(_l["config"]["frames_per_batch"])
md = [f"# {tag}: rocprofv3 evidence (MI355X, gfx950), collected by `profiles/collect_r05.sh {tag}`, summarised by `profiles/summarize_r05.py`.",
      f"bench.py = 26 x 6 person model, 640x480, batches of {BF} frames.", ""]
for name, what in (("batch8", f"batches of {BF} frames, one launch chain at a time, eager launches (bench.py --legs batchseq --graph 0 --inflight 1): the unit `roofline` is quoted on"),
                   ("seq", "single frames, one at a time (bench.py --legs seq --batch 1 --graph 0 --inflight 1): `roofline_single_frame`"),
                   ("seq1080", "1920x1080 single frames, one at a time")):
    ks, cj = os.path.join(src, f"{name}_kernel_stats.csv"), os.path.join(src, f"{name}_chains.json")
    if not (os.path.exists(ks) and os.path.exists(cj)):
        continue
    oname = f"batch{BF}" if name == "batch8" else name            # (the collector's directory label stays "batch8"; the committed files carry the batch size)
    shutil.copy(ks, os.path.join(HERE, f"{tag}_kernel_stats_{oname}.csv"))
    ch = json.load(open(cj))
    line = last_json_line(os.path.join(src, {"batch8": "trace8.json", "seq": "traceseq.json", "seq1080": "trace1080.json"}[name]))
    g = max(ch["groups"], key=lambda x: x["k_root_grid_threads"]) if ch["groups"] else None
    if name == "batch8" and g:
        ch["frames_per_launch"] = BF
        ch["benched_chain"] = g
        if line:
            ch["bench_line_of_this_run"] = {"roofline.launch_ms (HIP events, same process, under the profiler)": line["roofline"]["launch_ms"],
                                            "roofline.frac": line["roofline"]["frac"], "stage_ms_per_frame_batched": line.get("stage_ms_per_frame_batched")}
    json.dump(ch, open(os.path.join(HERE, f"{tag}_chains_{oname}.json"), "w"), indent=1)
    md += [f"## kernel trace: {what}", "| kernel | grid (threads) | calls | avg us | min us | max us | total ms |", "|---|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(ks)):
        if float(r["total_ms"]) < 0.02:
            continue
        md.append(f"| `{r['kernel']}` | {r['grid_threads']} | {r['calls']} | {r['avg_us']} | {r['min_us']} | {r['max_us']} | {r['total_ms']} |")
    if g:
        md += ["", f"dp_min launch chain ({g['chains']} chains after {ch['skipped_leading_chains']} skipped, {g['launches_per_chain']:.0f} launches each): "
               f"**sum of kernel durations {g['sum_of_kernel_durations_ms']:.4f} ms**, first start -> last end {g['span_first_start_to_last_end_ms']:.4f} ms per chain:"]
        for k, v in g["per_kernel"].items():
            md.append(f"* `{k}`: {v['launches_per_chain']:.0f} x {v['avg_us']:.2f} us = {v['ms_per_chain']:.4f} ms")
        if line:
            md.append(f"* the bench line of the same run: `roofline.launch_ms` {line['roofline']['launch_ms']} (HIP events around the stage), frac {line['roofline']['frac']}")
    md.append("")
# ---- HBM traffic of the batch chains
fr, wr = counter_rows("pmc8_FETCH_SIZE"), counter_rows("pmc8_WRITE_SIZE")
if fr and wr:
    cf, cw = chains_of(fr, "FETCH_SIZE"), chains_of(wr, "WRITE_SIZE")
    gb, gs = max(cf), min(cf)
    fbb, wbb = cf[gb][1] * 1e3 / cf[gb][0], cw[gb][1] * 1e3 / cw[gb][0]
    tj = {"round": tag, "source": f"profiles/{tag}_rocprof_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over bench.py --legs batchseq --graph 0 --inflight 1)",
          "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md HBM section); WRITE_SIZE uncalibrated, taken as is",
          "batch": {"frames_per_launch": BF, "chains": cf[gb][0], "fetch_bytes_raw": fbb, "write_bytes": wbb,
                    "hbm_bytes_per_launch_corrected": 2 * fbb + wbb, "hbm_bytes_per_frame_corrected": (2 * fbb + wbb) / BF}}
    if gs != gb:   # the one single-frame chain of the threshold pick
        fs, ws = cf[gs][1] * 1e3 / cf[gs][0], cw[gs][1] * 1e3 / cw[gs][0]
        tj.update({"fetch_bytes_raw": fs, "write_bytes": ws, "hbm_bytes_per_frame_corrected": 2 * fs + ws, "single_frame_chains": cf[gs][0]})
    json.dump(tj, open(os.path.join(HERE, "traffic_dp.json"), "w"), indent=1)
    md += ["## HBM traffic of the dp_min chain (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass, KB; FETCH x2 per MI355X_MICROARCH.md)",
           f"batch of {BF}: fetch {fbb / 1e6:.1f} MB raw (x2 = {2 * fbb / 1e6:.1f}) + write {wbb / 1e6:.1f} MB = {(2 * fbb + wbb) / 1e6:.1f} MB per chain = "
           f"**{(2 * fbb + wbb) / BF / 1e6:.1f} MB per frame** ({cf[gb][0]} chains)" +
           (f"; the single-frame chain of the same run: {tj['hbm_bytes_per_frame_corrected'] / 1e6:.1f} MB" if gs != gb else ""), ""]
    per = {}
    for rows, c in ((fr, "FETCH_SIZE"), (wr, "WRITE_SIZE")):
        for r in rows:
            if r["Counter_Name"] == c:
                a = per.setdefault((short(r["Kernel_Name"]), int(r["Grid_Size"])), {}).setdefault(c, [0, 0.0])
                a[0] += 1; a[1] += float(r["Counter_Value"])
    md += ["| kernel | grid | calls | FETCH KB/call (raw) | WRITE KB/call |", "|---|---|---|---|---|"]
    for (k, g), v in sorted(per.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", [0, 0])[1]):
        if not k.startswith("k_"):
            continue
        f_, w_ = v.get("FETCH_SIZE", [1, 0.0]), v.get("WRITE_SIZE", [1, 0.0])
        md.append(f"| `{k}` | {g} | {f_[0]} | {f_[1] / max(f_[0], 1):.1f} | {w_[1] / max(w_[0], 1):.1f} |")
    md.append("")
# ---- SQ counters (one pass each: sq8, and round 5's sq8b / sq8c)
sq = {}
for r in counter_rows("sq8") + [r for r in counter_rows("sq8b") if r["Counter_Name"] in ("SQ_THREAD_CYCLES_VALU", "SQ_INSTS_SALU")] + \
        [r for r in counter_rows("sq8c") if r["Counter_Name"] not in ("SQ_BUSY_CYCLES",)]:
    k = short(r["Kernel_Name"])
    if not k.startswith("k_"):
        continue
    a = sq.setdefault((k, int(r["Grid_Size"])), {}).setdefault(r["Counter_Name"], [0, 0.0])
    a[0] += 1; a[1] += float(r["Counter_Value"])
if sq:
    cs = sorted({c for k in sq for c in sq[k]})
    md += [f"## SQ counters per launch (one rocprofv3 --pmc pass, eight SQ counters; batches of {BF}, one chain at a time; *_CYCLES / ACTIVE / WAIT count quad-cycles summed over all waves)",
           "| kernel | grid | calls | " + " | ".join(c.replace("SQ_", "") for c in cs) + " | ACTIVE_INST_ANY / WAVE_CYCLES | LDS conflict cycles per LDS inst | active lanes per VALU inst (THREAD_CYCLES_VALU / ACTIVE_INST_VALU, of 64) |", "|---|---|---|" + "---|" * (len(cs) + 3)]
    tot_valu = {}
    for (k, g), v in sorted(sq.items()):
        m = {c: v[c][1] / v[c][0] for c in v}
        n = max(v[c][0] for c in v)
        wc, li = m.get("SQ_WAVE_CYCLES", 0), m.get("SQ_INSTS_LDS", 0)
        md.append(f"| `{k}` | {g} | {n} | " + " | ".join(f"{m[c]:.3g}" if c in m else "-" for c in cs) +
                  f" | {m.get('SQ_ACTIVE_INST_ANY', 0) / wc if wc else 0:.3f} | {m.get('SQ_LDS_BANK_CONFLICT', 0) / li if li else 0:.2f} | "
                  f"{m['SQ_THREAD_CYCLES_VALU'] / m['SQ_ACTIVE_INST_VALU'] if m.get('SQ_ACTIVE_INST_VALU') and 'SQ_THREAD_CYCLES_VALU' in m else float('nan'):.1f} |")
        if k.startswith(("k_dt_pass", "k_root")) and "SQ_INSTS_VALU" in v:
            tot_valu[g] = tot_valu.get(g, 0.0) + v["SQ_INSTS_VALU"][1]
    md.append("")
for t_ in ("batch_stages.txt",):
    p = os.path.join(src, t_)
    if os.path.exists(p) and os.path.getsize(p):
        md += [f"## {t_}", "```", open(p).read().strip(), "```", ""]
for b in ("bench_n1.json", "bench_n1_driverflags.json", "bench_n1_b1.json", "bench_n1_f64.json"):
    p = os.path.join(src, b)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(HERE, f"{tag}_{b}"))
        md += [f"## {b}", "```", open(p).read().strip(), "```", ""]
open(os.path.join(HERE, f"{tag}_rocprof_summary.md"), "w").write("\n".join(md) + "\n")
print("wrote", os.path.join(HERE, f"{tag}_rocprof_summary.md"))
