#!/bin/bash
# r04 session 26: does the 256-lane / 40 KB block geometry (4 wavefronts per SIMD instead of 3) hold at other frame sizes below 640x480, and
# where between 640x480 and 1920x1080 does it turn into a loss?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s26
export PBD_LIBRARY=$PWD/partsbaseddetector_amd/libpbd_hip_tune.so
run() {  # name, extra bench args, env...
  name=$1; shift; args=$1; shift
  env "$@" timeout 300 python bench.py --steps 60 --legs timed,batchseq,seq --warmup 5 --no-cpu-baseline $args > gpurun_out/r04s26/bench_$name.json 2> gpurun_out/r04s26/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r04s26/bench_$name.json').read().strip().splitlines()[-1])
    print('$name:', d['value'], 'batched dp_min', d['stage_ms_per_frame_batched']['dp_min'], 'seq dp_min', d['stage_ms_sequential']['dp_min'])
except Exception as e:
    print('$name: ERR', e)
PY
}
for sz in "320 240" "480 360" "800 600" "1024 768" "1280 720"; do
  set -- $sz
  run ${1}x${2}_default "--width $1 --height $2" X=1
  run ${1}x${2}_all256_40 "--width $1 --height $2" PBD_DT_NT=256 PBD_DT_NT_X=256 PBD_DT_BUDGET_KB=40 PBD_DT_BUDGET_X_KB=40
done
