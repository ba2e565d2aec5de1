#!/bin/bash
# session 35: full GPU suite on the final code + refreshed bench lines of the r03c collection (traffic_dp.json now holds the batch-of-8 chain)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=$REPO/gpurun_out/r03c; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 > $OUT/pytest_all.log 2>&1; echo "pytest rc=$?" > $OUT/summary.txt
tail -3 $OUT/pytest_all.log | cut -c1-200 >> $OUT/summary.txt
python bench.py --steps 300 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_n1_driverflags.json 2>> $OUT/bench_n1.err
python bench.py --steps 50 --dtype f64 > $OUT/bench_n1_f64.json 2>> $OUT/bench_n1.err
cat $OUT/summary.txt
python - <<'PY'
import json
for f in ('bench_n1','bench_n1_driverflags','bench_n1_f64'):
    d=json.loads(open(f'/root/repo/gpurun_out/r03c/{f}.json').read().strip().splitlines()[-1])
    print(f, d['value'], d['value_incl_h2d'], d['value_single_frame_calls'], d['roofline']['frac'], d['roofline'].get('traffic'), d['pdf']['frac'])
PY
