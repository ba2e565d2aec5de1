#!/bin/bash
# r04 session 13 (tuning build): filter-bank occupancy variants after the K-loop change (20 = default: two n-tiles at 3 waves per SIMD; 25 = two
# n-tiles at 4 waves per SIMD (26 spilled registers); 26 / 21 = one n-tile at 4 / 3 waves per SIMD), and the DT block budget at 1920x1080
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s13
export PBD_LIBRARY=$PWD/partsbaseddetector_amd/libpbd_hip_tune.so
for v in 20 25 26 21; do
  PBD_MFMA_VARIANT=$v timeout 200 python bench.py --legs timed,batchseq,seq --steps 40 --no-prewarm --warmup 5 > gpurun_out/r04s13/conv_$v.json 2> gpurun_out/r04s13/conv_$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04s13/conv_$v.json').read().strip().splitlines()[-1])
print('variant $v: value', d['value'], 'pdf batched', d['stage_ms_per_frame_batched']['pdf'], 'alone', d['stage_ms_sequential']['pdf'], 'dp batched', d['stage_ms_per_frame_batched']['dp_min'])
PY
done
timeout 300 python -m pytest tests -m gpu -q -x -k "tuning_variants" > gpurun_out/r04s13/pytest_variants.log 2>&1; tail -2 gpurun_out/r04s13/pytest_variants.log
for b in 25 32 40 56; do
  PBD_DT_BUDGET_KB=$b timeout 200 python bench.py --legs batchseq,seq --width 1920 --height 1080 --batch 2 --inflight 2 --no-prewarm --warmup 2 > gpurun_out/r04s13/dt1080_$b.json 2> gpurun_out/r04s13/dt1080_$b.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04s13/dt1080_$b.json').read().strip().splitlines()[-1])
print('1080p budget $b KB: dp_min batched', d['stage_ms_per_frame_batched']['dp_min'], 'alone', d['stage_ms_sequential']['dp_min'])
PY
done
