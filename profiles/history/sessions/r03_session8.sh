#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03h
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 > $OUT/pytest.log 2>&1
echo "pytest rc=$?" > $OUT/summary.txt; tail -6 $OUT/pytest.log >> $OUT/summary.txt
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_sequential']; print(d['value'], d['value_incl_h2d'], 'pdf', s['pdf'], 'dp', s['dp_min'], 'hog', s['hog'], 'pyr', s['image_pyramid'], 'pdfTF', d['pdf']['TFLOP/s'])"; }
tp() { echo "$1: $(python bench.py --steps 300 --no-cpu-baseline $2 2>/dev/null | line)" >> $OUT/summary.txt; }
tp "default" ""
tp "default again" ""
tp "inflight 1" "--inflight 1 --steps 100"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o run -- python $REPO/bench.py --graph 0 --no-prewarm --steps 4 --warmup 2 --inflight 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
python - <<'PY' >> $OUT/summary.txt
import csv, glob, collections
for f in glob.glob('/root/repo/gpurun_out/r03h/pmc_write/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'WRITE_SIZE':
            k = r['Kernel_Name'][:40]; acc[k][0] += 1; acc[k][1] += float(r['Counter_Value'])
    for k, (n, v) in sorted(acc.items(), key=lambda x: -x[1][1])[:8]:
        print(f"WRITE_SIZE {k}: calls {n} KB/call {v / n:.1f}")
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
cat $OUT/summary.txt
