#!/bin/bash
# r04 session 31: handles in flight x frames per batch with the round's final kernels (the optimum of round 3 was 3 x 8)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s31
for cfg in "3 8" "4 8" "2 8" "3 12" "3 16" "2 16" "4 4" "3 8"; do
  set -- $cfg
  timeout 200 python bench.py --steps $((1200 / $2)) --warmup 5 --legs timed --inflight $1 --batch $2 > gpurun_out/r04s31/bench_$1x$2.json 2> gpurun_out/r04s31/bench_$1x$2.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r04s31/bench_$1x$2.json').read().strip().splitlines()[-1])
    print('inflight $1 batch $2:', d['value'], d['ms_per_step'])
except Exception as e:
    print('inflight $1 batch $2: ERR', e)
PY
done
