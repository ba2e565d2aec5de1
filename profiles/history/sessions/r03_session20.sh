#!/bin/bash
# session 20: phase sums of the persistent filter bank (probe build), 1 / 2 / 3 workgroups per CU
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03s
mkdir -p $OUT
cd $REPO
for v in 19 10 11; do PBD_MFMA_VARIANT=$v python tests/tools_conv_glds_probe.py 2>>$OUT/err.log | tail -1 >> $OUT/summary.txt; done
export PBD_LIBRARY=$REPO/partsbaseddetector_amd/libpbd_hip_tune.so
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'pdf seq', d['stage_ms_sequential']['pdf'])"; }
PBD_MFMA_VARIANT=19 python bench.py --steps 30 --no-cpu-baseline --inflight 1 --batch 1 2>>$OUT/err.log | line >> $OUT/summary.txt
cat $OUT/summary.txt
