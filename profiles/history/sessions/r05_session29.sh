#!/bin/bash
# r05 session 29: two cheap knobs at the final default (3 x 16): hardware queues 4 / 8 / 16, thin DT launches (tuning build), interleaved twice
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s29; mkdir -p $O
T=$GRAFT_REPO_ROOT/partsbaseddetector_amd/libpbd_hip_tune.so
run() { # name, env...
  n=$1; shift
  env "$@" timeout 100 python bench.py --legs timed --steps 40 --warmup 4 > $O/$n.json 2> $O/$n.err
  python - $n <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.loads(open(f'gpurun_out/r05s29/{n}.json').read().strip().splitlines()[-1]); print(n, 'value', round(d['value'], 1))
except Exception as e:
    print(n, 'ERR', e)
PY
}
for rep in 1 2; do
  run q8_$rep GPU_MAX_HW_QUEUES=8
  run q4_$rep GPU_MAX_HW_QUEUES=4
  run q16_$rep GPU_MAX_HW_QUEUES=16
  run thin_$rep PBD_LIBRARY=$T PBD_DT_THIN=1
  run tune0_$rep PBD_LIBRARY=$T
done
