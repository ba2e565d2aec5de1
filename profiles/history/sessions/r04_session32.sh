#!/bin/bash
# r04 session 32: single frames: thin rounds (2 parts, 1 part: 520 / 260 blocks for 1024 slots) with proportionally smaller blocks
# (PBD_DT_THIN=1, tuning build; minimum budgets 10 / 16 / 24 KB) against the default
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s32
export PBD_LIBRARY=$PWD/partsbaseddetector_amd/libpbd_hip_tune.so
run() {
  name=$1; shift
  env "$@" timeout 200 python bench.py --steps 60 --legs batchseq,seq --warmup 3 --no-cpu-baseline > gpurun_out/r04s32/bench_$name.json 2> gpurun_out/r04s32/bench_$name.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04s32/bench_$name.json').read().strip().splitlines()[-1])
print('$name: batched dp_min', d['stage_ms_per_frame_batched']['dp_min'], 'seq dp_min', d['stage_ms_sequential']['dp_min'])
PY
}
run default X=1
run thin10 PBD_DT_THIN=1
run thin16 PBD_DT_THIN=1 PBD_DT_THIN_MIN_KB=16
run thin24 PBD_DT_THIN=1 PBD_DT_THIN_MIN_KB=24
