#!/bin/bash
# session 36: plain DT loader with unpredicated LDS stores: parity subset + timing
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03af; mkdir -p $OUT; cd $REPO
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 -k "dt or dp or detect or fold" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" > $OUT/summary.txt
tail -2 $OUT/pytest.log | cut -c1-200 >> $OUT/summary.txt
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'dp seq', d['stage_ms_sequential']['dp_min'], 'dp batched', (d.get('stage_ms_per_frame_batched') or {}).get('dp_min'))"; }
for i in 1 2; do echo "run $i: $(python bench.py --steps 60 --no-cpu-baseline 2>>$OUT/err.log | line)" >> $OUT/summary.txt; done
cat $OUT/summary.txt
