#!/bin/bash
# round 3, GPU session 5: full GPU suite (new adaptor tests), conv stagger sweep, frames-in-flight sweep
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03e
mkdir -p $OUT
cd $REPO
timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 > $OUT/pytest_all.log 2>&1
echo "pytest all rc=$?" > $OUT/summary.txt
tail -12 $OUT/pytest_all.log >> $OUT/summary.txt
export PBD_LIBRARY=$REPO/partsbaseddetector_amd/libpbd_hip_tune.so
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_sequential']; print(d['value'], d['value_incl_h2d'], 'pdf', s['pdf'], 'dp', s['dp_min'], 'hog', s['hog'], 'pyr', s['image_pyramid'])"; }
tp() { echo "$1: $(python bench.py --steps 200 --no-cpu-baseline $2 2>/dev/null | line)" >> $OUT/sweep.txt; }
for st in 0 4 8 12 16 24; do PBD_DP_MODE=1 PBD_CONV_STAGGER_US=$st tp "legacy stagger ${st}us" ""; done
PBD_CONV_STAGGER_US=12 tp "fold stagger 12us" ""
for s in 2 3 6 8; do PBD_DP_MODE=1 tp "legacy inflight $s" "--inflight $s"; done
cat $OUT/sweep.txt >> $OUT/summary.txt
cat $OUT/summary.txt
