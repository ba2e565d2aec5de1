#!/bin/bash
# session 38: variant 18 (LDS-DMA staged, one unit per workgroup) against the default at 3 handles x batches of 8
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03ag; mkdir -p $OUT; cd $REPO
export PBD_LIBRARY=$REPO/partsbaseddetector_amd/libpbd_hip_tune.so
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['value_single_frame_calls'], 'pdf seq', d['stage_ms_sequential']['pdf'], 'pdf batched', (d.get('stage_ms_per_frame_batched') or {}).get('pdf'))"; }
for v in 20 18 20 18 20 18; do echo "variant $v: $(PBD_MFMA_VARIANT=$v python bench.py --steps 60 --no-cpu-baseline 2>>$OUT/err.log | line)" >> $OUT/summary.txt; done
cat $OUT/summary.txt
