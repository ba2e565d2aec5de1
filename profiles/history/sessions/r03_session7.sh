#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03g
mkdir -p $OUT
cd $REPO
export PBD_LIBRARY=$REPO/partsbaseddetector_amd/libpbd_hip_tune.so
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_sequential']; print(d['value'], d['value_incl_h2d'], 'pdf', s['pdf'], 'dp', s['dp_min'])"; }
tp() { echo "$1: $(python bench.py --steps 300 --no-cpu-baseline $2 2>/dev/null | line)" >> $OUT/summary.txt; }
PBD_DP_MODE=1 tp "legacy" ""
tp "fold auto" ""
PBD_DT_NO_RESIDENT=1 tp "fold no-resident (25k)" ""
PBD_DT_BUDGET_X_KB=28 tp "fold x 28k" ""
PBD_DT_BUDGET_X_KB=32 tp "fold x 32k" ""
PBD_DP_MODE=1 tp "legacy again" ""
PBD_DT_NO_RESIDENT=1 tp "fold no-resident again" ""
cat $OUT/summary.txt
