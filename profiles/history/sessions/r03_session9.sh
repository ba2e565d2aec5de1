#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03i
mkdir -p $OUT
cd $REPO
timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 -k "compact or 1080p or group or bench_lines or adaptors or foreign" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" > $OUT/summary.txt; tail -8 $OUT/pytest.log >> $OUT/summary.txt
grep "1920x1080 person" $OUT/pytest.log >> $OUT/summary.txt
python tests/tools_dt_trace.py 640 480 2 > $OUT/trace_l2.txt 2>&1
python tests/tools_dt_trace.py 640 480 3 > $OUT/trace_l3.txt 2>&1
grep "^launch" $OUT/trace_l2.txt | cut -c1-175 >> $OUT/summary.txt
python bench.py --steps 100 --dtype f64 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('f64', d['value'], d['stage_ms_sequential'])" >> $OUT/summary.txt
python bench.py --steps 20 --width 1920 --height 1080 --inflight 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('1080p', d['value'], d['stage_ms_sequential'])" >> $OUT/summary.txt
cat $OUT/summary.txt
