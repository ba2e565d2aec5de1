#!/bin/bash
# session 18: the round's rocprofv3 collection (profiles/collect.sh r03a) + short-run S x B choice at the driver's flags
bash profiles/collect.sh r03a > gpurun_out/collect_r03a.log 2>&1
OUT=$PWD/gpurun_out/r03a
for sb in "4 3" "3 4" "3 8" "2 8" "4 2"; do set -- $sb
  python bench.py --steps 20 --warmup 5 --inflight $1 --batch $2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps 20: inflight $1 batch $2:', d['value'], d['value_incl_h2d'], d['value_single_frame_calls'], d['roofline']['frac'], d['roofline']['launch_ms'])" >> $OUT/sweep_short.txt
done
cat $OUT/sweep_sb.txt $OUT/sweep_short.txt $OUT/batch_stages.txt
tail -3 gpurun_out/collect_r03a.log
