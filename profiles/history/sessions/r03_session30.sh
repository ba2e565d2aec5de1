#!/bin/bash
# session 30: filter-bank occupancy cap (LDS request) with frames in flight: does leaving LDS to the DT blocks pay?
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03ab
mkdir -p $OUT
cd $REPO
export PBD_LIBRARY=$REPO/partsbaseddetector_amd/libpbd_hip_tune.so
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'pdf seq', d['stage_ms_sequential']['pdf'], 'pdf batched', (d.get('stage_ms_per_frame_batched') or {}).get('pdf'))"; }
tp() { echo "$1: $(python bench.py --steps ${3:-40} --no-cpu-baseline $2 2>>$OUT/err.log | line)" >> $OUT/summary.txt; }
tp "default" ""
for kb in 33 41 54 81; do PBD_CONV_LDS_KB=$kb tp "conv lds request $kb KB" ""; done
PBD_MFMA_VARIANT=22 tp "variant 22" ""
PBD_MFMA_VARIANT=22 PBD_CONV_LDS_KB=54 tp "variant 22 + 54 KB" ""
cat $OUT/summary.txt
