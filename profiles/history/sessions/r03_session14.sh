#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03n
mkdir -p $OUT
cd $REPO
export PBD_LIBRARY=$REPO/partsbaseddetector_amd/libpbd_hip_tune.so
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['value_incl_h2d'], 'ms/step', d['ms_per_step'])"; }
tp() { echo "$1: $(python bench.py --steps $3 --no-cpu-baseline $2 2>>$OUT/err.log | line)" >> $OUT/summary.txt; }
tp "S3 B4 default" "--inflight 3 --batch 4" 100
PBD_DP_MODE=1 tp "S3 B4 three-kernel" "--inflight 3 --batch 4" 100
PBD_DT_BUDGET_KB=28 tp "S3 B4 budget 28k" "--inflight 3 --batch 4" 100
PBD_DT_BUDGET_KB=32 tp "S3 B4 budget 32k" "--inflight 3 --batch 4" 100
PBD_DT_BUDGET_KB=22 tp "S3 B4 budget 22k" "--inflight 3 --batch 4" 100
PBD_DT_NT_X=192 PBD_DT_BUDGET_X_KB=30 tp "S3 B4 x nt192 30k" "--inflight 3 --batch 4" 100
tp "S3 B8" "--inflight 3 --batch 8" 50
tp "S4 B6" "--inflight 4 --batch 6" 60
tp "S2 B6" "--inflight 2 --batch 6" 60
tp "S5 B3" "--inflight 5 --batch 3" 100
cat $OUT/summary.txt
