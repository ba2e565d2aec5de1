#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03l
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -q -x -k "batch or detect_exact or pyramid or resize or pyrdown or stagewise or compact" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" > $OUT/summary.txt; tail -6 $OUT/pytest.log >> $OUT/summary.txt
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_sequential']; print(d['value'], d['value_incl_h2d'], 'ms/step', d['ms_per_step'])"; }
tp() { echo "$1: $(python bench.py --steps $3 --no-cpu-baseline $2 2>>$OUT/err.log | line)" >> $OUT/summary.txt; }
tp "S4 B1" "--inflight 4 --batch 1" 300
tp "S2 B2" "--inflight 2 --batch 2" 150
tp "S2 B4" "--inflight 2 --batch 4" 100
tp "S1 B4" "--inflight 1 --batch 4" 100
tp "S3 B4" "--inflight 3 --batch 4" 100
tp "S2 B8" "--inflight 2 --batch 8" 60
tp "S1 B8" "--inflight 1 --batch 8" 60
tp "S4 B2" "--inflight 4 --batch 2" 150
cat $OUT/summary.txt; tail -5 $OUT/err.log
