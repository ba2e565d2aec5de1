#!/bin/bash
# r05 session 2: (a) whole GPU suite, no -x (HOG now emits the split parts; compact plan shares them with the DT pointer planes);
# (b) split-bank variants 0 / 1 (pinned block schedule, 2 / 4 wavefronts), 4 / 5 (loads dealt between the MFMAs), 3 (compiler's);
# (c) co-scheduling sweep: DT block budget x bank variant x handles in flight (tuning build); (d) kernel trace + matrix-pipe counters
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s2; mkdir -p $O
TUNE=$PWD/partsbaseddetector_amd/libpbd_hip_tune.so
one() {  # <label> <env...> -- <bench args...>
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env PBD_LIBRARY=$TUNE "${envs[@]}" timeout 300 python bench.py "$@" 2> $O/$label.err > $O/$label.json
  python - $O/$label.json "$label" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    sb=d.get('stage_ms_per_frame_batched') or {}
    print(f"{sys.argv[2]:34s} value {d['value']}  pdf {sb.get('pdf')} dp {sb.get('dp_min')} total {sb.get('total')}  roof {d['roofline']['frac']}", flush=True)
except Exception as e: print(sys.argv[2], 'failed', e, flush=True)
PY
}
for v in 0 1 4 5 3; do one var$v PBD_SPLIT_VARIANT=$v -- --steps 100 --conv split --legs timed,batchseq; done | tee $O/variants.txt
for v in 1 5 3; do
  for kb in 40 52 32; do
    for s in 3 4; do one co_v${v}_kb${kb}_s$s PBD_SPLIT_VARIANT=$v PBD_DT_BUDGET_KB=$kb PBD_DT_BUDGET_X_KB=$kb -- --steps 100 --conv split --legs timed --inflight $s; done
  done
done | tee $O/cosched.txt
one co_v5_nt128 PBD_SPLIT_VARIANT=5 PBD_DT_NT=128 PBD_DT_NT_X=128 PBD_DT_BUDGET_KB=25 PBD_DT_BUDGET_X_KB=25 -- --steps 100 --conv split --legs timed,batchseq | tee -a $O/cosched.txt
timeout 300 python bench.py --steps 100 --legs timed,mfma32 > $O/bench_mfma32.json 2> $O/bench_mfma32.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05s2/bench_mfma32.json').read().strip().splitlines()[-1])
print('value', d['value'], 'mfma32', d.get('value_fp32_mfma'), d.get('value_fp32_mfma_frame_ms'))
PY
bash profiles/collect_r05.sh r05a trace8 sq3 > $O/collect.log 2>&1; tail -3 $O/collect.log
cat gpurun_out/r05a/batch8_kernel_stats.csv | head -30
timeout 1700 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
tail -40 $O/pytest_all.log
