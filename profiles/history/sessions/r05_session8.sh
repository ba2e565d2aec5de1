#!/bin/bash
# r05 session 8: smoke(); the split-bank fuzz (two seeds) and the random-detector fuzz on the final code; HOG tile size 8 against 16 (tuning build)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s8; mkdir -p $O
TUNE=$PWD/partsbaseddetector_amd/libpbd_hip_tune.so
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
timeout 200 python tests/tools_fuzz_split.py 60 11 > $O/fuzz_split.log 2>&1; echo "rc=$?" >> $O/fuzz_split.log
timeout 200 python tests/tools_fuzz_split.py 60 12 >> $O/fuzz_split.log 2>&1; echo "rc=$?" >> $O/fuzz_split.log; cat $O/fuzz_split.log
timeout 300 python tests/tools_fuzz_detect.py 90 21 > $O/fuzz_detect.log 2>&1; echo "rc=$?" >> $O/fuzz_detect.log; tail -2 $O/fuzz_detect.log
one() {  # <label> <env...> -- <bench args...>
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py "$@" 2> $O/$label.err > $O/$label.json
  python - $O/$label.json "$label" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    sb=d.get('stage_ms_per_frame_batched') or {}; ss=d.get('stage_ms_sequential') or {}
    print(f"{sys.argv[2]:20s} value {d['value']}  hog {sb.get('hog')} pdf {sb.get('pdf')} dp {sb.get('dp_min')} total {sb.get('total')} | alone hog {ss.get('hog')} total {ss.get('total')}", flush=True)
except Exception as e: print(sys.argv[2], 'failed', e, flush=True)
PY
}
for i in 1 2; do
  one tc16_$i PBD_LIBRARY=$TUNE -- --steps 30 --legs timed,batchseq,seq
  one tc8_$i PBD_LIBRARY=$TUNE PBD_HOG_TC=8 -- --steps 30 --legs timed,batchseq,seq
done | tee $O/hog_tc.txt
timeout 600 python -m pytest tests -m gpu -q -x -k "fuzz_split or hog or pyramid" > $O/pytest_sub.log 2>&1; echo "rc=$?" >> $O/pytest_sub.log; tail -3 $O/pytest_sub.log
