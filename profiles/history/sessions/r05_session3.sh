#!/bin/bash
# r05 session 3: new default split-bank variant (4 wavefronts, loads dealt between the MFMAs): run-to-run spread; k_dt_pass at 96 registers
# (tune5 build) against the product's 104-109 (A/B, alternating); the driver-flags line with the fp32-MFMA child leg; T = double DT geometry; suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s3; mkdir -p $O
TUNE=$PWD/partsbaseddetector_amd/libpbd_hip_tune.so; TUNE5=$PWD/partsbaseddetector_amd/libpbd_hip_tune5.so
one() {  # <label> <env...> -- <bench args...>
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py "$@" 2> $O/$label.err > $O/$label.json
  python - $O/$label.json "$label" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    sb=d.get('stage_ms_per_frame_batched') or {}
    print(f"{sys.argv[2]:28s} value {d['value']}  pdf {sb.get('pdf')} dp {sb.get('dp_min')} hog {sb.get('hog')} total {sb.get('total')}  roof {d['roofline']['frac']}", flush=True)
except Exception as e: print(sys.argv[2], 'failed', e, flush=True)
PY
}
for i in 1 2 3; do one default_$i X=1 -- --steps 100 --legs timed,batchseq; done | tee $O/spread.txt
for i in 1 2; do
  one tune_$i PBD_LIBRARY=$TUNE -- --steps 100 --legs timed,batchseq
  one tune5_$i PBD_LIBRARY=$TUNE5 -- --steps 100 --legs timed,batchseq
done | tee $O/dtwpe.txt
PBD_LIBRARY=$TUNE5 timeout 600 python -m pytest tests -m gpu -q -x -k "dp_min or dt2d or detect_exact or fuzz or detect_random" > $O/pytest_tune5.log 2>&1; echo "rc=$?" >> $O/pytest_tune5.log; tail -3 $O/pytest_tune5.log
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_driverflags.json 2> $O/bench_driverflags.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05s3/bench_driverflags.json').read().strip().splitlines()[-1])
print('driverflags value', d['value'], 'mfma32', d.get('value_fp32_mfma'), 'incl_h2d', d['value_incl_h2d'], 'single', d['value_single_frame_calls'], 'seq', d['sequential']['latency_ms'], 'roof', d['roofline']['frac'], d['roofline_single_frame']['frac'], d['stage_ms_per_frame_batched'], d['stage_ms_sequential'], d['pdf'], 'cpu', d.get('cpu_baseline',{}).get('value'))
PY
one f64_default PBD_LIBRARY=$TUNE -- --steps 50 --dtype f64 --legs timed,seq,batchseq | tee $O/f64.txt
one f64_nt128_40k PBD_LIBRARY=$TUNE PBD_DT_NT=128 PBD_DT_NT_X=128 PBD_DT_BUDGET_KB=40 PBD_DT_BUDGET_X_KB=40 -- --steps 50 --dtype f64 --legs timed,seq,batchseq | tee -a $O/f64.txt
timeout 1700 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
tail -8 $O/pytest_all.log
