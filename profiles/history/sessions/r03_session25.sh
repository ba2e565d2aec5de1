#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03x; mkdir -p $OUT; cd $REPO
PBD_DT_BF_MAXLEN=256 python tests/tools_dt_trace.py 640 480 1 > $OUT/trace_bf_l1.txt 2>$OUT/err.log
PBD_DT_BF_MAXLEN=256 python tests/tools_dt_trace.py 640 480 0 > $OUT/trace_bf_l0.txt 2>>$OUT/err.log
head -3 $OUT/trace_bf_l1.txt | cut -c1-250
grep -v "^launch" $OUT/trace_bf_l1.txt | head -32
grep -v "^launch" $OUT/trace_bf_l0.txt | sed -n 2,12p
