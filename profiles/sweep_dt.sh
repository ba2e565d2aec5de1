# A/B of the DT planner knobs with the tuning build (make -C partsbaseddetector_amd/csrc tune): bash profiles/sweep_dt.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export PBD_LIBRARY=$REPO/partsbaseddetector_amd/libpbd_hip_tune.so
for nt in ${NTS:-64 128}; do for seg in ${SEGS:-0 12 16 20 24 32}; do for kb in ${KBS:-20}; do
  export PBD_DT_NT=$nt PBD_DT_SEG=$seg PBD_DT_BUDGET_B=$kb   # bytes
  a=$(python $REPO/bench.py ${DTYPE:+--dtype $DTYPE} --steps 30 --inflight 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['stage_ms_sequential']['dp_min'])")
  b=$(python $REPO/bench.py ${DTYPE:+--dtype $DTYPE} --steps 200 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])")
  echo "nt $nt seg $seg kb $kb: dp_min $a ms, $b frames/s"
done; done; done
if [ -n "$HD" ]; then for kb in ${KBS:-20}; do for nt in ${NTS:-64 128}; do
  export PBD_DT_NT=$nt PBD_DT_SEG=0 PBD_DT_BUDGET_KB=$kb
  a=$(python $REPO/bench.py --width 1920 --height 1080 --steps 10 --inflight 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['stage_ms_sequential']['dp_min'], d['value'])")
  echo "1080p nt $nt kb $kb: dp_min ms, frames/s: $a"
done; done; fi
