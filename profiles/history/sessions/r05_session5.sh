#!/bin/bash
# r05 session 5: fold loader fetching the next child's planes under this child's arithmetic (tune5 build) against the product kernels (A/B, alternating),
# its parity subset; handles in flight x frames per batch with the round's kernels
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s5; mkdir -p $O
TUNE=$PWD/partsbaseddetector_amd/libpbd_hip_tune.so; TUNE5=$PWD/partsbaseddetector_amd/libpbd_hip_tune5.so
one() {  # <label> <env...> -- <bench args...>
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py "$@" 2> $O/$label.err > $O/$label.json
  python - $O/$label.json "$label" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    sb=d.get('stage_ms_per_frame_batched') or {}; ss=d.get('stage_ms_sequential') or {}
    print(f"{sys.argv[2]:28s} value {d['value']}  ms/step {d['ms_per_step']} pdf {sb.get('pdf')} dp {sb.get('dp_min')} total {sb.get('total')} | alone dp {ss.get('dp_min')} total {ss.get('total')}  roof {d['roofline']['frac']}", flush=True)
except Exception as e: print(sys.argv[2], 'failed', e, flush=True)
PY
}
for i in 1 2 3; do
  one tune_$i PBD_LIBRARY=$TUNE -- --steps 40 --legs timed,batchseq,seq
  one prefetch_$i PBD_LIBRARY=$TUNE5 -- --steps 40 --legs timed,batchseq,seq
done | tee $O/fold_prefetch.txt
PBD_LIBRARY=$TUNE5 timeout 600 python -m pytest tests -m gpu -q -x -k "dp_min or detect_exact or detect_random or fold or config5 or timed_configuration" > $O/pytest_tune5.log 2>&1; echo "rc=$?" >> $O/pytest_tune5.log; tail -3 $O/pytest_tune5.log
for sb in "3 8" "4 8" "3 12" "2 12" "4 6"; do set -- $sb
  one sb_$1_$2 X=1 -- --steps 30 --legs timed --inflight $1 --batch $2
done | tee $O/sweep_sb.txt
