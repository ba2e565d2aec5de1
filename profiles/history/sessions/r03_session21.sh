#!/bin/bash
# session 21: persistent filter bank with 16-byte B loads: parity, phase sums, stage time and throughput
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03t
mkdir -p $OUT
cd $REPO
for v in 19 10 11; do PBD_MFMA_VARIANT=$v python tests/tools_conv_glds_probe.py 2>>$OUT/err.log | tail -1 >> $OUT/summary.txt; done
export PBD_LIBRARY=$REPO/partsbaseddetector_amd/libpbd_hip_tune.so
PBD_MFMA_VARIANT=10 timeout 600 python -m pytest tests -m gpu -q -x -k "pdf or mfma" > $OUT/pytest_v10.log 2>&1; echo "pytest v10 rc=$?" >> $OUT/summary.txt
tail -2 $OUT/pytest_v10.log >> $OUT/summary.txt
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['value_incl_h2d'], 'single', d['value_single_frame_calls'], 'pdf seq', d['stage_ms_sequential']['pdf'], 'pdf batched', (d.get('stage_ms_per_frame_batched') or {}).get('pdf'))"; }
tp() { echo "$1: $(python bench.py --steps $3 --no-cpu-baseline $2 2>>$OUT/err.log | line)" >> $OUT/summary.txt; }
for v in 5 10 11; do
  PBD_MFMA_VARIANT=$v tp "variant $v S4 B3" "--inflight 4 --batch 3" 100
done
PBD_MFMA_VARIANT=10 tp "variant 10 S3 B4" "--inflight 3 --batch 4" 80
cat $OUT/summary.txt
