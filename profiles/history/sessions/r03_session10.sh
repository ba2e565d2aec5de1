#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03j
mkdir -p $OUT
cd $REPO
timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 > $OUT/pytest.log 2>&1
echo "pytest rc=$?" > $OUT/summary.txt; tail -8 $OUT/pytest.log >> $OUT/summary.txt
python tests/tools_dt_trace.py 640 480 2 > $OUT/trace_l2.txt 2>&1
grep "^launch" $OUT/trace_l2.txt | cut -c1-175 >> $OUT/summary.txt
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_sequential']; print(d['value'], d['value_incl_h2d'], 'pdf', s['pdf'], 'dp', s['dp_min'], 'seq lat', d['sequential']['latency_ms']['median'])"; }
tp() { echo "$1: $(python bench.py --steps 300 --no-cpu-baseline $2 2>/dev/null | line)" >> $OUT/summary.txt; }
tp "default" ""
tp "default again" ""
export PBD_LIBRARY=$REPO/partsbaseddetector_amd/libpbd_hip_tune.so
PBD_DP_MODE=1 tp "three-kernel" ""
cat $OUT/summary.txt
