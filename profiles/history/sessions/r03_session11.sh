#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03k
mkdir -p $OUT
cd $REPO
export PBD_LIBRARY=$REPO/partsbaseddetector_amd/libpbd_hip_tune.so
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_sequential']; print(d['value'], d['value_incl_h2d'], 'pdf', s['pdf'], 'dp', s['dp_min'])"; }
tp() { echo "$1: $(python bench.py --steps 300 --no-cpu-baseline $2 2>/dev/null | line)" >> $OUT/summary.txt; }
tp "default" ""
PBD_DT_NT_X=192 PBD_DT_BUDGET_X_KB=30 tp "x nt192 30k" ""
PBD_DT_NT_X=192 PBD_DT_BUDGET_X_KB=31 tp "x nt192 31k" ""
PBD_DT_NT_X=256 PBD_DT_BUDGET_X_KB=31 tp "x nt256 31k" ""
PBD_DT_NT_X=256 PBD_DT_BUDGET_X_KB=40 tp "x nt256 40k" ""
PBD_DT_NT_X=192 PBD_DT_BUDGET_X_KB=25 tp "x nt192 25k" ""
PBD_DT_REVERSE=1 tp "reverse order" ""
tp "default again" ""
cat $OUT/summary.txt
