#!/bin/bash
# round 3, GPU session 1: parity of the fold structure, first numbers, thin-round geometry sweep
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03a
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -q --maxfail=15 -x -k "dp_min or dt2d or detect_exact or smoke or stagewise" > $OUT/pytest_dp.log 2>&1
echo "pytest dp rc=$?" >> $OUT/summary.txt
tail -5 $OUT/pytest_dp.log >> $OUT/summary.txt
timeout 1200 python -m pytest tests -m gpu -q --maxfail=15 > $OUT/pytest_all.log 2>&1
echo "pytest all rc=$?" >> $OUT/summary.txt
tail -15 $OUT/pytest_all.log >> $OUT/summary.txt
python bench.py --steps 200 --no-cpu-baseline > $OUT/bench_tp.json 2> $OUT/bench_tp.err
python bench.py --steps 60 --inflight 1 --no-cpu-baseline > $OUT/bench_seq.json 2> $OUT/bench_seq.err
python - <<'PY' >> $OUT/summary.txt
import json
for f in ("bench_tp","bench_seq"):
    try:
        d=json.load(open(f"/root/repo/gpurun_out/r03a/{f}.json"))
        print(f, d["value"], d["stage_ms_sequential"], d["roofline"]["frac"])
    except Exception as e: print(f, "ERR", e)
PY
export PBD_LIBRARY=$REPO/partsbaseddetector_amd/libpbd_hip_tune.so
run() {
  a=$(python bench.py --steps 40 --inflight 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['stage_ms_sequential']['dp_min'])")
  b=$(python bench.py --steps 200 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])")
  echo "$1: dp_min $a ms, $b frames/s" >> $OUT/sweep.txt
}
PBD_DT_THIN=0 run "thin0"
for mk in 8 10 13; do PBD_DT_THIN=1 PBD_DT_THIN_MIN_KB=$mk run "thin1 min${mk}k"; done
for kb in 20 30; do PBD_DT_BUDGET_KB=$kb run "budget${kb}k"; done
PBD_DT_NT=256 PBD_DT_BUDGET_KB=32 run "nt256 budget32k"; PBD_DT_NT=256 PBD_DT_BUDGET_KB=40 run "nt256 budget40k"
cat $OUT/sweep.txt >> $OUT/summary.txt
cat $OUT/summary.txt
