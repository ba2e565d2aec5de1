#!/bin/bash
# r04 session 24: geometry sweep, second pass: fold x blocks of 256 lanes at 36-46 KB; plain blocks of 256 lanes at 40 / 50 KB with them
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s24
export PBD_LIBRARY=$PWD/partsbaseddetector_amd/libpbd_hip_tune.so
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 150 --legs timed,batchseq,seq --warmup 5 --no-cpu-baseline > gpurun_out/r04s24/bench_$name.json 2> gpurun_out/r04s24/bench_$name.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04s24/bench_$name.json').read().strip().splitlines()[-1])
print('$name:', d['value'], 'batched dp_min', d['stage_ms_per_frame_batched']['dp_min'], 'seq dp_min', d['stage_ms_sequential']['dp_min'])
PY
}
run default X=1
run x256_36 PBD_DT_NT_X=256 PBD_DT_BUDGET_X_KB=36
run x256_38 PBD_DT_NT_X=256 PBD_DT_BUDGET_X_KB=38
run x256_40 PBD_DT_NT_X=256 PBD_DT_BUDGET_X_KB=40
run x256_42 PBD_DT_NT_X=256 PBD_DT_BUDGET_X_KB=42
run x256_46 PBD_DT_NT_X=256 PBD_DT_BUDGET_X_KB=46
run all256_40 PBD_DT_NT=256 PBD_DT_NT_X=256 PBD_DT_BUDGET_KB=40 PBD_DT_BUDGET_X_KB=40
run all256_50 PBD_DT_NT=256 PBD_DT_NT_X=256 PBD_DT_BUDGET_KB=50 PBD_DT_BUDGET_X_KB=50
run y256_40_x256_40_plain128 PBD_DT_NT_X=256 PBD_DT_BUDGET_X_KB=40 PBD_DT_SEG=12
run default2 X=1
