#!/bin/bash
# round 3, GPU session 4: fold loader without scalar-load chains, targeted validation, flat plain loader, full-lane lpb,
# residency-aware budgets: parity + traces + A/B
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03d
mkdir -p $OUT
cd $REPO
timeout 600 python -m pytest tests -m gpu -q -x -k "dp_min or dt2d or detect_exact or stagewise or f64" > $OUT/pytest_dp.log 2>&1
echo "pytest dp rc=$?" > $OUT/summary.txt
tail -3 $OUT/pytest_dp.log >> $OUT/summary.txt
python tests/tools_dt_trace.py 640 480 2 > $OUT/trace_fold_l2.txt 2>&1
python tests/tools_dt_trace.py 640 480 0 > $OUT/trace_fold_l0.txt 2>&1
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
runtp "fold auto"
PBD_DP_MODE=1 runtp "legacy auto"
PBD_DT_NO_RESIDENT=1 run "fold no-resident"
for kb in 25 32; do PBD_DT_BUDGET_X_KB=$kb run "fold x budget ${kb}k"; done
for kb in 22 28; do PBD_DT_BUDGET_KB=$kb run "fold base ${kb}k"; PBD_DP_MODE=1 PBD_DT_BUDGET_KB=$kb run "legacy base ${kb}k"; done
PBD_DT_SEG=20 run "fold seg20"
PBD_DP_MODE=1 PBD_DT_SEG=20 run "legacy seg20"
cat $OUT/sweep.txt >> $OUT/summary.txt
grep "^launch" $OUT/trace_fold_l2.txt | cut -c1-170 >> $OUT/summary.txt
grep "^launch" $OUT/trace_legacy_l2.txt | cut -c1-170 >> $OUT/summary.txt
cat $OUT/summary.txt
