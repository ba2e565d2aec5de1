#!/bin/bash
# round 3, GPU session 2: where does the fold x pass spend its time?  block traces (probe build) + A/B (tuning build)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03b
mkdir -p $OUT
cd $REPO
timeout 600 python -m pytest tests -m gpu -q -x -k "dp_min or dt2d or detect_exact or stagewise or f64" > $OUT/pytest_dp.log 2>&1
echo "pytest dp rc=$?" > $OUT/summary.txt
tail -3 $OUT/pytest_dp.log >> $OUT/summary.txt
python tests/tools_dt_trace.py 640 480 2 > $OUT/trace_fold_l2.txt 2>&1
python tests/tools_dt_trace.py 640 480 0 > $OUT/trace_fold_l0.txt 2>&1
python tests/tools_dt_trace.py 640 480 3 > $OUT/trace_fold_l3.txt 2>&1
PBD_DP_MODE=1 python tests/tools_dt_trace.py 640 480 2 > $OUT/trace_legacy_l2.txt 2>&1
export PBD_LIBRARY=$REPO/partsbaseddetector_amd/libpbd_hip_tune.so
run() {
  a=$(python bench.py --steps 30 --inflight 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['stage_ms_sequential']['dp_min'])")
  echo "$1: dp_min $a ms" >> $OUT/sweep.txt
}
runtp() {
  b=$(python bench.py --steps 200 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_sequential']['dp_min'])")
  echo "$1: frames/s, dp_min: $b" >> $OUT/sweep.txt
}
runtp "fold 25k"
PBD_DP_MODE=1 runtp "legacy 25k"
for kb in 22 28 30 32 36; do PBD_DT_BUDGET_KB=$kb run "fold budget${kb}k"; done
for kb in 22 28 32; do PBD_DP_MODE=1 PBD_DT_BUDGET_KB=$kb run "legacy budget${kb}k"; done
PBD_DT_NT=192 PBD_DT_BUDGET_KB=30 run "fold nt192 30k"
PBD_DT_NT=192 PBD_DT_BUDGET_KB=36 run "fold nt192 36k"
cat $OUT/sweep.txt >> $OUT/summary.txt
head -30 $OUT/trace_fold_l2.txt >> $OUT/summary.txt
cat $OUT/summary.txt
