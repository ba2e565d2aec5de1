#!/usr/bin/env python
"""Reduce a rocprofv3 --kernel-trace CSV ON THE GPU BOX (the per-dispatch trace is too large to travel back) into

  <out>_kernel_stats.csv   per (kernel, grid size): calls, average / min / max / total duration
  <out>_chains.json        the dp_min launch chains of the run (k_dt_pass launches up to and including a k_root launch, in
                           dispatch order), grouped by the k_root grid size — a chain over a batch of B frames has B times the
                           grid of a single-frame chain, so the chains of the benched unit are told apart from the one
                           single-frame chain bench.py's threshold pick launches — with, per group: chains, launches per chain,
                           sum of the kernels' average durations, mean first-start -> last-end span, and the same per kernel name.

    python profiles/reduce_trace.py <dir holding *kernel_trace.csv> <out prefix> [skip_chains]

skip_chains: leading chains of every group left out of the averages (warm-up launches: plan, LDS opt-in, cold caches).
No third-party imports: runs with the box's bare python.
"""
import csv
import glob
import json
import os
import sys


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def col(row, *names):
    for n in names:
        if n in row and row[n] != "":
            return row[n]
    return "0"


def main():
    src, out = sys.argv[1], sys.argv[2]
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    files = glob.glob(os.path.join(src, "*kernel_trace.csv")) + glob.glob(os.path.join(src, "*", "*kernel_trace.csv"))
    if not files:
        raise SystemExit(f"no *kernel_trace.csv under {src}")
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            gx, gy, gz = (int(col(r, f"Grid_Size_{a}", "Grid_Size" if a == "X" else "_")) or 1 for a in "XYZ")
            rows.append((int(col(r, "Dispatch_Id")), short(r["Kernel_Name"]), gx * gy * gz, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows.sort()
    # ---- per (kernel, grid) statistics — of the dispatches AFTER the warm-up: everything up to and including the (skip + 1)-th
    # k_root launch of the run (bench.py's threshold pick + `skip` chains: plan, LDS opt-ins, first-call code loads) is left out for
    # EVERY kernel (round 4 skipped the warm-up chains for the dp_min sums only: the filter bank's average then held one 26 ms first call)
    roots = [i for i, k, _, _, _ in rows if k.startswith("k_root")]
    cutoff = roots[skip] if len(roots) > skip + 1 else -1
    st = {}
    for i, k, g, t0, t1 in rows:
        if i <= cutoff:
            continue
        a = st.setdefault((k, g), [0, 0, 1 << 62, 0])
        d = t1 - t0
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    with open(out + "_kernel_stats.csv", "w") as f:
        f.write("kernel,grid_threads,calls,avg_us,min_us,max_us,total_ms\n")
        for (k, g), a in sorted(st.items(), key=lambda kv: -kv[1][1]):
            f.write(f"\"{k}\",{g},{a[0]},{a[1] / a[0] / 1e3:.3f},{a[2] / 1e3:.3f},{a[3] / 1e3:.3f},{a[1] / 1e6:.4f}\n")
    # ---- dp_min chains
    groups, cur = {}, []
    for _, k, g, t0, t1 in rows:
        if k.startswith(("k_dt_pass", "k_reduce")):
            cur.append((k, t0, t1))
        elif k.startswith("k_root"):
            cur.append((k, t0, t1))
            groups.setdefault(g, []).append(cur)
            cur = []
    res = {"source": [os.path.basename(f) for f in files], "skipped_leading_chains": skip,
           "kernel_stats_skip_dispatches_up_to_id": cutoff, "groups": []}
    for g in sorted(groups):
        ch = groups[g][skip:] if len(groups[g]) > skip else groups[g]
        per = {}
        for c in ch:
            for k, t0, t1 in c:
                a = per.setdefault(k, [0, 0])
                a[0] += 1; a[1] += t1 - t0
        n = len(ch)
        res["groups"].append({
            "k_root_grid_threads": g, "chains": n, "launches_per_chain": sum(len(c) for c in ch) / n,
            "sum_of_kernel_durations_ms": sum(t1 - t0 for c in ch for _, t0, t1 in c) / n / 1e6,
            "span_first_start_to_last_end_ms": sum(c[-1][2] - c[0][1] for c in ch) / n / 1e6,
            "per_kernel": {k: {"launches_per_chain": a[0] / n, "avg_us": a[1] / a[0] / 1e3, "ms_per_chain": a[1] / n / 1e6} for k, a in sorted(per.items())}})
    json.dump(res, open(out + "_chains.json", "w"), indent=1)
    print(json.dumps(res["groups"], indent=1))


if __name__ == "__main__":
    main()
