#!/bin/bash
# session 29: DT block geometry in batch mode (tuning build): budget / lanes / segment sweep at 3 handles x batches of 8, and larger batches
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03aa
mkdir -p $OUT
cd $REPO
export PBD_LIBRARY=$REPO/partsbaseddetector_amd/libpbd_hip_tune.so
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'dp seq', d['stage_ms_sequential']['dp_min'], 'dp batched', (d.get('stage_ms_per_frame_batched') or {}).get('dp_min'))"; }
tp() { echo "$1: $(python bench.py --steps ${3:-40} --no-cpu-baseline $2 2>>$OUT/err.log | line)" >> $OUT/summary.txt; }
tp "default S3 B8" ""
PBD_DT_BUDGET_KB=20 tp "budget 20 KB" ""
PBD_DT_BUDGET_KB=30 tp "budget 30 KB" ""
PBD_DT_BUDGET_KB=36 tp "budget 36 KB" ""
PBD_DT_BUDGET_KB=48 tp "budget 48 KB" ""
PBD_DT_NO_RESIDENT=1 tp "no resident search" ""
PBD_DT_SEG=12 tp "seg 12" ""
PBD_DT_SEG=32 tp "seg 32" ""
PBD_DT_NT=128 tp "nt 128" ""
PBD_DT_NT_X=128 tp "nt_x 128" ""
tp "S3 B16" "--inflight 3 --batch 16" 20
tp "S2 B16" "--inflight 2 --batch 16" 20
tp "S4 B8" "--inflight 4 --batch 8" 30
cat $OUT/summary.txt
