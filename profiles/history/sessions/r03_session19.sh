#!/bin/bash
# session 19: persistent double-buffered filter bank (k_conv_glds, PBD_MFMA_VARIANT 10 = 2 workgroups/CU, 11 = 3): parity + timing
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03r
mkdir -p $OUT
cd $REPO
export PBD_LIBRARY=$REPO/partsbaseddetector_amd/libpbd_hip_tune.so
PBD_MFMA_VARIANT=10 timeout 600 python -m pytest tests -m gpu -q -x -k "pdf or mfma" > $OUT/pytest_v10.log 2>&1; echo "pytest v10 rc=$?" > $OUT/summary.txt
tail -3 $OUT/pytest_v10.log >> $OUT/summary.txt
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['value_incl_h2d'], 'single', d['value_single_frame_calls'], 'pdf seq', d['stage_ms_sequential']['pdf'], 'pdf batched', (d.get('stage_ms_per_frame_batched') or {}).get('pdf'))"; }
tp() { echo "$1: $(python bench.py --steps $3 --no-cpu-baseline $2 2>>$OUT/err.log | line)" >> $OUT/summary.txt; }
for v in 5 10 11; do
  PBD_MFMA_VARIANT=$v tp "variant $v S4 B3" "--inflight 4 --batch 3" 100
done
for v in 10 11; do
  PBD_MFMA_VARIANT=$v tp "variant $v S3 B4" "--inflight 3 --batch 4" 80
done
cat $OUT/summary.txt
