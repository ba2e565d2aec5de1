#!/bin/bash
# r05 session 24: larger batches (timed leg only), interleaved with the 3 x 16 point of session 23
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s24; mkdir -p $O
for rep in 1 2; do
  for sb in "3 16" "3 24" "3 32" "2 32" "2 24" "4 16"; do
    set -- $sb
    timeout 120 python bench.py --legs timed --steps 40 --warmup 4 --inflight $1 --batch $2 > $O/b_$1x$2_$rep.json 2> $O/b_$1x$2_$rep.err
    python - $1 $2 $rep <<'PY'
import json, sys
s, b, r = sys.argv[1:]
try:
    d = json.loads(open(f'gpurun_out/r05s24/b_{s}x{b}_{r}.json').read().strip().splitlines()[-1])
    print(f'rep {r}  {s} x {b}: value {d["value"]:.1f}  ms_per_step {d["ms_per_step"]:.3f}')
except Exception as e:
    print(f'rep {r} {s} x {b}: ERR {e}')
PY
  done
done
