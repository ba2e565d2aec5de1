set -u
O=gpurun_out/r06_s14; mkdir -p $O
for rep in 1 2; do
for cfg in "3 16" "4 16" "3 20" "3 24" "4 12" "2 24" "5 12"; do
  set -- $cfg
  python bench.py --steps $((960 / ($1 * $2))) --warmup 3 --legs timed --inflight $1 --batch $2 > $O/i$1_b$2_$rep.json 2>> $O/err.log
  python - $O/i$1_b$2_$rep.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['config']['frames_per_step_per_gpu'])
PY
done
done
