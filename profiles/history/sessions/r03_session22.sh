#!/bin/bash
# session 22: 16-byte B loads in k_conv_mfma16 (variants 20-22) vs the persistent kernel (10): parity + stage time + throughput
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03u
mkdir -p $OUT
cd $REPO
export PBD_LIBRARY=$REPO/partsbaseddetector_amd/libpbd_hip_tune.so
PBD_MFMA_VARIANT=20 timeout 600 python -m pytest tests -m gpu -q -x -k "pdf or mfma" > $OUT/pytest_v20.log 2>&1; echo "pytest v20 rc=$?" >> $OUT/summary.txt
tail -2 $OUT/pytest_v20.log >> $OUT/summary.txt
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['value_incl_h2d'], 'single', d['value_single_frame_calls'], 'pdf seq', d['stage_ms_sequential']['pdf'], 'pdf batched', (d.get('stage_ms_per_frame_batched') or {}).get('pdf'))"; }
tp() { echo "$1: $(python bench.py --steps $3 --no-cpu-baseline $2 2>>$OUT/err.log | line)" >> $OUT/summary.txt; }
for v in 5 20 21 22 10; do
  PBD_MFMA_VARIANT=$v tp "variant $v S4 B3" "--inflight 4 --batch 3" 100
done
for v in 20 22; do
  PBD_MFMA_VARIANT=$v tp "variant $v S3 B4" "--inflight 3 --batch 4" 80
done
cat $OUT/summary.txt
