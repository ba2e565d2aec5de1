#!/bin/bash
# r04 session 25: geometry sweep, third pass: plain blocks of 256 lanes at 36-46 KB with fold x blocks of 256 lanes at 40 KB; 1920x1080 for
# the two best against the default
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s25
export PBD_LIBRARY=$PWD/partsbaseddetector_amd/libpbd_hip_tune.so
run() {  # name, extra bench args, env...
  name=$1; shift; args=$1; shift
  env "$@" timeout 300 python bench.py --steps 150 --legs timed,batchseq,seq --warmup 5 --no-cpu-baseline $args > gpurun_out/r04s25/bench_$name.json 2> gpurun_out/r04s25/bench_$name.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04s25/bench_$name.json').read().strip().splitlines()[-1])
print('$name:', d['value'], 'batched dp_min', d['stage_ms_per_frame_batched']['dp_min'], 'seq dp_min', d['stage_ms_sequential']['dp_min'])
PY
}
run default "" X=1
for kb in 36 38 40 42 44 46; do run p256_${kb}_x256_40 "" PBD_DT_NT=256 PBD_DT_NT_X=256 PBD_DT_BUDGET_KB=$kb PBD_DT_BUDGET_X_KB=40; done
run p128_25_x256_40 "" PBD_DT_NT_X=256 PBD_DT_BUDGET_X_KB=40
HD="--width 1920 --height 1080 --steps 30"
run hd_default "$HD" X=1
run hd_all256_40 "$HD" PBD_DT_NT=256 PBD_DT_NT_X=256 PBD_DT_BUDGET_KB=40 PBD_DT_BUDGET_X_KB=40
run hd_x256_40 "$HD" PBD_DT_NT_X=256 PBD_DT_BUDGET_X_KB=40
