#!/bin/bash
# r04 session 6: (1) filter bank: K loop instantiated per M-tile count (no branch per MFMA) + any kh x kw: parity tests; (2) A/B of the fold
# loader variants on the dp_min stage: default (bias block by vector loads) / v1 (scalar bias loads) / v2 (commit 49fe431: before the fold / setup diet)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s6
timeout 900 python -m pytest tests -m gpu -q -x -k "pdf or filter_size or 7x7 or config5 or timed_configuration or tuning" > gpurun_out/r04s6/pytest_conv.log 2>&1; echo "rc=$?" >> gpurun_out/r04s6/pytest_conv.log
tail -4 gpurun_out/r04s6/pytest_conv.log
for v in default ab_v1 ab_v2; do
  if [ $v = default ]; then unset PBD_LIBRARY; else export PBD_LIBRARY=$PWD/partsbaseddetector_amd/libpbd_hip_$v.so; fi
  timeout 300 python bench.py --legs batchseq,seq --no-prewarm --warmup 3 > gpurun_out/r04s6/bench_$v.json 2> gpurun_out/r04s6/bench_$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04s6/bench_$v.json').read().strip().splitlines()[-1])
print('$v', 'batched', d['stage_ms_per_frame_batched'], 'seq', d['stage_ms_sequential'])
PY
done
unset PBD_LIBRARY
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04s6/bench_driverflags.json 2> gpurun_out/r04s6/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04s6/bench_driverflags.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','value_incl_h2d','value_single_frame_calls')}, d['roofline']['frac'], d['roofline']['launch_ms'], d['pdf'])
PY
