#!/usr/bin/env python
"""Reduce a rocprofv3 --kernel-trace CSV of bench.py's TIMED leg (three batches in flight on three streams, hipGraph replay) into the concurrency it shows:
how much of the time two or more kernels of different batches are running at once, and in particular how much of the split-product filter bank's
run time a distance-transform launch of another batch is running beside it (VERDICT r04 #2 asks for exactly that trace).  Run ON THE BOX (the trace is
large), prints one JSON object.

    rocprofv3 --kernel-trace --output-format csv -d <dir> -o run -- python bench.py --steps 30 --legs timed
    python profiles/overlap_trace.py <dir> [skip_fraction]

An interval = first wavefront start .. last wavefront end of a dispatch: two overlapping intervals are kernels whose workgroups were on the chip in the
same span (not necessarily on the same CU).  The first `skip_fraction` of the dispatches (pre-warm, warm-up: default 0.5) is left out.  No third-party imports."""
import csv
import glob
import json
import os
import sys


def cls(name):
    n = name.split("(")[0].replace("void ", "").strip()
    if n.startswith("k_conv") or n.startswith("k_feat_split"):
        return "pdf"
    if n.startswith(("k_dt_pass", "k_root", "k_reduce")):
        return "dp_min"
    if n.startswith("k_hog"):
        return "hog"
    if n.startswith(("k_resize", "k_pyrdown")):
        return "pyramid"
    if n.startswith("k_backtrack"):
        return "argmin"
    return "other"


def main():
    src = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    files = glob.glob(os.path.join(src, "*kernel_trace.csv")) + glob.glob(os.path.join(src, "*", "*kernel_trace.csv"))
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), cls(r["Kernel_Name"]), r.get("Queue_Id", r.get("Stream_Id", "?"))))
    rows.sort()
    rows = rows[int(len(rows) * skip):]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    ev = []
    for s, e, c, q in rows:
        ev.append((s, 1, c)); ev.append((e, -1, c))
    ev.sort()
    active = {}
    last = t0
    busy = {"any": 0, "2+": 0, "3+": 0, "pdf": 0, "dp_min": 0, "pdf&dp_min": 0, "pdf&hog": 0, "dp_min&hog": 0, "dp_min&dp_min": 0}
    for t, d, c in ev:
        dt = t - last
        if dt > 0:
            n = sum(active.values())
            if n >= 1: busy["any"] += dt
            if n >= 2: busy["2+"] += dt
            if n >= 3: busy["3+"] += dt
            if active.get("pdf", 0): busy["pdf"] += dt
            if active.get("dp_min", 0): busy["dp_min"] += dt
            if active.get("pdf", 0) and active.get("dp_min", 0): busy["pdf&dp_min"] += dt
            if active.get("pdf", 0) and active.get("hog", 0): busy["pdf&hog"] += dt
            if active.get("dp_min", 0) and active.get("hog", 0): busy["dp_min&hog"] += dt
            if active.get("dp_min", 0) >= 2: busy["dp_min&dp_min"] += dt
        active[c] = active.get(c, 0) + d
        last = t
    wall = t1 - t0
    ksum = {}
    for s, e, c, q in rows:
        ksum[c] = ksum.get(c, 0) + (e - s)
    tot = sum(ksum.values())
    out = {"dispatches": len(rows), "queues": len({r[3] for r in rows}), "wall_ms": wall / 1e6,
           "sum_of_kernel_durations_ms": tot / 1e6, "sum_over_wall": tot / wall,
           "chip_busy_fraction_of_wall": busy["any"] / wall, "two_or_more_kernels_fraction_of_wall": busy["2+"] / wall,
           "three_or_more_kernels_fraction_of_wall": busy["3+"] / wall,
           "pdf_running_fraction_of_wall": busy["pdf"] / wall, "dp_min_running_fraction_of_wall": busy["dp_min"] / wall,
           "fraction_of_pdf_time_with_a_dp_min_launch_running": busy["pdf&dp_min"] / max(busy["pdf"], 1),
           "fraction_of_pdf_time_with_hog_running": busy["pdf&hog"] / max(busy["pdf"], 1),
           "fraction_of_dp_min_time_with_another_dp_min_chain_running": busy["dp_min&dp_min"] / max(busy["dp_min"], 1),
           "kernel_time_share": {k: v / tot for k, v in sorted(ksum.items())}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
