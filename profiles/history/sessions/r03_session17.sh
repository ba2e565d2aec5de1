#!/bin/bash
# session 17: full GPU suite with the new defaults + default bench lines
OUT=$PWD/gpurun_out/r03q; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 > $OUT/pytest_all.log 2>&1; echo "pytest rc=$?" > $OUT/summary.txt
tail -5 $OUT/pytest_all.log >> $OUT/summary.txt
( time python bench.py --steps 20 --warmup 5 ) > $OUT/bench_driver.json 2> $OUT/bench_driver.err
python bench.py --steps 200 --no-cpu-baseline > $OUT/bench_200.json 2>> $OUT/bench_driver.err
python - <<'PY' >> $OUT/summary.txt
import json
for f in ('bench_driver','bench_200'):
    try:
        d=json.loads(open(f'/root/repo/gpurun_out/r03q/{f}.json').read().strip().splitlines()[-1])
        print(f, d['value'], d.get('value_single_frame_calls'), d.get('value_incl_h2d'), d['ms_per_step'], d['roofline'], d.get('stage_ms'))
    except Exception as e: print(f, 'ERR', e)
PY
grep real $OUT/bench_driver.err >> $OUT/summary.txt
cat $OUT/summary.txt
