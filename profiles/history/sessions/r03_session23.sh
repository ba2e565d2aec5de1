#!/bin/bash
# session 23: phase split of the DT blocks (probe build): launch 0 (x pass, fold loader) and launch 1 (y pass, plain loader)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03v; mkdir -p $OUT; cd $REPO
python tests/tools_dt_trace.py 640 480 0 > $OUT/trace_l0.txt 2>$OUT/err.log
python tests/tools_dt_trace.py 640 480 1 > $OUT/trace_l1.txt 2>>$OUT/err.log
head -22 $OUT/trace_l0.txt | cut -c1-250
grep -v "^launch" $OUT/trace_l0.txt | head -30
grep -v "^launch" $OUT/trace_l1.txt | head -30
