#!/bin/bash
# r05 session 23: handles in flight x frames per batch with the final tree, interleaved (timed leg only, 100 steps of S batches each)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s23; mkdir -p $O
for rep in 1 2 3; do
  for sb in "3 8" "3 12" "4 8" "3 16" "4 12"; do
    set -- $sb
    timeout 120 python bench.py --legs timed --steps 60 --warmup 5 --inflight $1 --batch $2 > $O/b_$1x$2_$rep.json 2> $O/b_$1x$2_$rep.err
    python - $1 $2 $rep <<'PY'
import json, sys
s, b, r = sys.argv[1:]
try:
    d = json.loads(open(f'gpurun_out/r05s23/b_{s}x{b}_{r}.json').read().strip().splitlines()[-1])
    print(f'rep {r}  {s} x {b}: value {d["value"]:.1f}  ms_per_step {d["ms_per_step"]:.3f}')
except Exception as e:
    print(f'rep {r} {s} x {b}: ERR {e}')
PY
  done
done
