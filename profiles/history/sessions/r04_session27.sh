#!/bin/bash
# r04 session 27: float DT blocks of 256 lanes / 40 KB while every line has byte links (the default now): parity tests, then A/B against the
# old geometry through the tuning build's knobs (PBD_DT_NT=128 PBD_DT_BUDGET_KB=25), 640x480 and 1280x720 (which must not change)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s27
timeout 900 python -m pytest tests -m gpu -q -x -k "dt or dp or detect or batch" > gpurun_out/r04s27/pytest_dt.log 2>&1; echo "rc=$?" >> gpurun_out/r04s27/pytest_dt.log
tail -4 gpurun_out/r04s27/pytest_dt.log
export PBD_LIBRARY=$PWD/partsbaseddetector_amd/libpbd_hip_tune.so
run() {  # name, extra bench args, env...
  name=$1; shift; args=$1; shift
  env "$@" timeout 300 python bench.py --steps 150 --legs timed,batchseq,seq --warmup 5 --no-cpu-baseline $args > gpurun_out/r04s27/bench_$name.json 2> gpurun_out/r04s27/bench_$name.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04s27/bench_$name.json').read().strip().splitlines()[-1])
print('$name:', d['value'], 'batched dp_min', d['stage_ms_per_frame_batched']['dp_min'], 'seq dp_min', d['stage_ms_sequential']['dp_min'], 'single-frame calls', d.get('value_single_frame_calls'))
PY
}
run new "" X=1
run old "" PBD_DT_NT=128 PBD_DT_BUDGET_KB=25
run new2 "" X=1
run old2 "" PBD_DT_NT=128 PBD_DT_BUDGET_KB=25
run 720p_new "--width 1280 --height 720 --steps 40" X=1
run 720p_old "--width 1280 --height 720 --steps 40" PBD_DT_NT=128 PBD_DT_BUDGET_KB=25
