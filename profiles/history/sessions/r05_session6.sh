#!/bin/bash
# r05 session 6: n-tile groups of four (4 + 1) and three (3 + 2) instead of five: 242 / 190 registers per wavefront -> two wavefronts per SIMD and room for
# two distance-transform wavefronts beside them; parity of the two forms; the bench line with the latency of a graph-replayed detect()
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s6; mkdir -p $O
TUNE=$PWD/partsbaseddetector_amd/libpbd_hip_tune.so
timeout 900 python -m pytest tests -m gpu -q -x -k "split_tuning" > $O/pytest_split.log 2>&1; echo "rc=$?" >> $O/pytest_split.log; tail -3 $O/pytest_split.log
one() {  # <label> <env...> -- <bench args...>
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py "$@" 2> $O/$label.err > $O/$label.json
  python - $O/$label.json "$label" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    sb=d.get('stage_ms_per_frame_batched') or {}
    print(f"{sys.argv[2]:28s} value {d['value']}  ms/step {d['ms_per_step']} pdf {sb.get('pdf')} dp {sb.get('dp_min')} hog {sb.get('hog')} total {sb.get('total')}  roof {d['roofline']['frac']}", flush=True)
except Exception as e: print(sys.argv[2], 'failed', e, flush=True)
PY
}
for i in 1 2 3; do
  one v0_$i PBD_LIBRARY=$TUNE PBD_SPLIT_VARIANT=0 -- --steps 40 --legs timed,batchseq
  one v7_$i PBD_LIBRARY=$TUNE PBD_SPLIT_VARIANT=7 -- --steps 40 --legs timed,batchseq
  one v8_$i PBD_LIBRARY=$TUNE PBD_SPLIT_VARIANT=8 -- --steps 40 --legs timed,batchseq
done | tee $O/ngroups.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_driverflags.json 2> $O/bench_driverflags.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05s6/bench_driverflags.json').read().strip().splitlines()[-1])
print('driverflags value', d['value'], 'ms/step', d['ms_per_step'], 'mfma32', d.get('value_fp32_mfma'), 'incl_h2d', d['value_incl_h2d'], 'single', d['value_single_frame_calls'], 'seq', d['sequential'], 'roof', d['roofline']['frac'], d['stage_ms_per_frame_batched'])
PY
