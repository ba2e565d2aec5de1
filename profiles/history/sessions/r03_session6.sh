#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03f
mkdir -p $OUT
cd $REPO
timeout 300 python -m pytest tests -m gpu -q -x -k "adaptors or foreign or demo" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" > $OUT/summary.txt; tail -4 $OUT/pytest.log >> $OUT/summary.txt
python tests/tools_stage_saturation.py 4 60 >> $OUT/summary.txt 2>&1
DP_MODE=2 python tests/tools_stage_saturation.py 4 60 >> $OUT/summary.txt 2>&1
python tests/tools_stage_saturation.py 1 60 >> $OUT/summary.txt 2>&1
export PBD_LIBRARY=$REPO/partsbaseddetector_amd/libpbd_hip_tune.so
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_sequential']; print(d['value'], d['value_incl_h2d'], 'pdf', s['pdf'], 'dp', s['dp_min'])"; }
tp() { echo "$1: $(python bench.py --steps 200 --no-cpu-baseline $2 2>/dev/null | line)" >> $OUT/summary.txt; }
for st in 0 6 12 20; do PBD_DP_MODE=1 PBD_CONV_STAGGER_US=$st tp "legacy stagger ${st}us" ""; done
cat $OUT/summary.txt
