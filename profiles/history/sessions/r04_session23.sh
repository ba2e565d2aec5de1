#!/bin/bash
# r04 session 23: fold x-pass block geometry (tuning build): lanes per block x LDS budget -> wavefronts per CU, lines per block (multiples of
# the 6 mixtures), blocks per launch (the 4-part rounds of a single frame: 1772 blocks for 1536 slots at the default)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s23
export PBD_LIBRARY=$PWD/partsbaseddetector_amd/libpbd_hip_tune.so
for cfg in "128 25" "192 30" "192 33" "192 37" "256 40" "256 50" "128 25"; do
  set -- $cfg
  PBD_DT_NT_X=$1 PBD_DT_BUDGET_X_KB=$2 timeout 300 python bench.py --steps 150 --legs timed,batchseq,seq --warmup 5 --no-cpu-baseline > gpurun_out/r04s23/bench_$1_$2.json 2> gpurun_out/r04s23/bench_$1_$2.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04s23/bench_$1_$2.json').read().strip().splitlines()[-1])
print('nt_x $1 budget_x $2 KB:', d['value'], 'batched dp_min', d['stage_ms_per_frame_batched']['dp_min'], 'seq dp_min', d['stage_ms_sequential']['dp_min'])
PY
done
