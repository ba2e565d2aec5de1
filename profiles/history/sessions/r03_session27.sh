#!/bin/bash
# session 27: single-buffer LDS-DMA filter bank (variant 18) vs the default (20) and the persistent one (10)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03z
mkdir -p $OUT
cd $REPO
export PBD_LIBRARY=$REPO/partsbaseddetector_amd/libpbd_hip_tune.so
PBD_MFMA_VARIANT=18 timeout 600 python -m pytest tests -m gpu -q -x -k "pdf or mfma" > $OUT/pytest_v18.log 2>&1; echo "pytest v18 rc=$?" >> $OUT/summary.txt
tail -2 $OUT/pytest_v18.log >> $OUT/summary.txt
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['value_incl_h2d'], 'single', d['value_single_frame_calls'], 'pdf seq', d['stage_ms_sequential']['pdf'], 'pdf batched', (d.get('stage_ms_per_frame_batched') or {}).get('pdf'))"; }
tp() { echo "$1: $(python bench.py --steps $3 --no-cpu-baseline $2 2>>$OUT/err.log | line)" >> $OUT/summary.txt; }
for v in 20 18 20 18; do
  PBD_MFMA_VARIANT=$v tp "variant $v S4 B3" "--inflight 4 --batch 3" 100
done
cat $OUT/summary.txt
