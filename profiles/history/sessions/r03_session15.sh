#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03p
mkdir -p $OUT
cd $REPO
export PBD_LIBRARY=$REPO/partsbaseddetector_amd/libpbd_hip_tune.so
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['value_incl_h2d'], 'ms/step', d['ms_per_step'], 'pdf', d['stage_ms_sequential']['pdf'])"; }
tp() { echo "$1: $(python bench.py --steps $3 --no-cpu-baseline $2 2>>$OUT/err.log | line)" >> $OUT/summary.txt; }
tp "S3 B4 default" "--inflight 3 --batch 4" 100
PBD_MFMA_VARIANT=4 tp "S3 B4 conv quarters (13.5 KB)" "--inflight 3 --batch 4" 100
PBD_MFMA_VARIANT=2 tp "S3 B4 conv halves 5 waves" "--inflight 3 --batch 4" 100
PBD_MFMA_VARIANT=1 tp "S3 B4 conv whole tile" "--inflight 3 --batch 4" 100
PBD_MFMA_VARIANT=5 tp "S3 B4 conv NTW2" "--inflight 3 --batch 4" 100
PBD_MFMA_VARIANT=4 tp "S4 B4 conv quarters" "--inflight 4 --batch 4" 100
cat $OUT/summary.txt
