#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03m
mkdir -p $OUT
cd $REPO
python tests/tools_batch_stages.py 1 2 4 8 > $OUT/summary.txt 2>&1
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['value_incl_h2d'], 'ms/step', d['ms_per_step'])"; }
tp() { echo "$1: $(python bench.py --steps $3 --no-cpu-baseline $2 2>>$OUT/err.log | line)" >> $OUT/summary.txt; }
tp "S4 B4" "--inflight 4 --batch 4" 100
tp "S3 B3" "--inflight 3 --batch 3" 120
tp "S3 B6" "--inflight 3 --batch 6" 80
tp "S6 B2" "--inflight 6 --batch 2" 150
tp "S3 B4" "--inflight 3 --batch 4" 100
tp "S4 B3" "--inflight 4 --batch 3" 120
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b4 -o run -- python $REPO/bench.py --graph 0 --no-prewarm --steps 10 --warmup 3 --inflight 1 --batch 4 --no-cpu-baseline > $OUT/stats_b4.log 2>&1
python - <<'PY' >> $OUT/summary.txt
import csv, glob
for f in glob.glob('/root/repo/gpurun_out/r03m/stats_b4/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print(r['Name'][:60], 'calls', r['Calls'], 'avg us', round(float(r['AverageNs'])/1e3,1), 'total ms', round(int(r['TotalDurationNs'])/1e6,2), '%', r['Percentage'])
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
cat $OUT/summary.txt
