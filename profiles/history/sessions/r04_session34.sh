#!/bin/bash
# r04 session 34: T = double: DT block geometry (default one wavefront / 20 KB = 8 blocks = 2 wavefronts per SIMD) against two wavefronts
# at 26 KB (6 blocks = 3 per SIMD: the register allocation's limit) and 40 KB (4 blocks = 2 per SIMD)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s34
export PBD_LIBRARY=$PWD/partsbaseddetector_amd/libpbd_hip_tune.so
run() {
  name=$1; shift
  env "$@" timeout 200 python bench.py --dtype f64 --steps 30 --legs batchseq,seq --warmup 3 --no-cpu-baseline > gpurun_out/r04s34/bench_$name.json 2> gpurun_out/r04s34/bench_$name.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04s34/bench_$name.json').read().strip().splitlines()[-1])
print('$name: batched dp_min', d['stage_ms_per_frame_batched']['dp_min'], 'seq dp_min', d['stage_ms_sequential']['dp_min'])
PY
}
run default X=1
run nt128_26 PBD_DT_NT=128 PBD_DT_BUDGET_KB=26
run nt128_40 PBD_DT_NT=128 PBD_DT_BUDGET_KB=40
