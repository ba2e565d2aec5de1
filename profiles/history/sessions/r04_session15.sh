#!/bin/bash
# r04 session 15 (probe build): per-block phase traces of the DT launches of one 640x480 person frame after the round's changes
# (launch 0 = round 0's plain x pass, launch 1 = its y pass, launch 2 = round 1's fold x pass)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s15
for l in 0 1 2; do timeout 200 python tests/tools_dt_trace.py 640 480 $l > gpurun_out/r04s15/trace_launch$l.txt 2> gpurun_out/r04s15/trace_launch$l.err; done
head -22 gpurun_out/r04s15/trace_launch0.txt; tail -12 gpurun_out/r04s15/trace_launch2.txt
