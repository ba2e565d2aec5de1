#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03q
mkdir -p $OUT
cd $REPO
export PBD_LIBRARY=$REPO/partsbaseddetector_amd/libpbd_hip_tune.so
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['value_incl_h2d'], 'ms/step', d['ms_per_step'], 'pdf', d['stage_ms_sequential']['pdf'])"; }
tp() { echo "$1: $(python bench.py --steps $3 --no-cpu-baseline $2 2>>$OUT/err.log | line)" >> $OUT/summary.txt; }
PBD_MFMA_VARIANT=5 tp "S3 B4 NTW2 (3 w/SIMD)" "--inflight 3 --batch 4" 100
PBD_MFMA_VARIANT=6 tp "S3 B4 NTW2 (2 w/SIMD alloc)" "--inflight 3 --batch 4" 100
PBD_MFMA_VARIANT=8 tp "S3 B4 NTW5 (2 w/SIMD)" "--inflight 3 --batch 4" 100
PBD_MFMA_VARIANT=9 tp "S3 B4 NTW5 (3 w/SIMD, spills)" "--inflight 3 --batch 4" 100
PBD_MFMA_VARIANT=5 tp "S4 B4 NTW2" "--inflight 4 --batch 4" 100
PBD_MFMA_VARIANT=5 tp "S4 B3 NTW2" "--inflight 4 --batch 3" 120
PBD_MFMA_VARIANT=5 tp "S4 B1 NTW2" "--inflight 4 --batch 1" 300
PBD_MFMA_VARIANT=5 tp "S3 B8 NTW2" "--inflight 3 --batch 8" 50
cat $OUT/summary.txt
