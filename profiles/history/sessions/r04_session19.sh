#!/bin/bash
# r04 session 19: read-out as a flat state machine (one event per iteration: an output or a step down the chain; the nested form costs the
# wavefront as many dependent LDS round trips per output as its busiest lane steps).  DT parity tests, then A/B: default (flat) / ab_nested
# (the same source with -DDT_READOUT_FLAT=0 = session 18's kernels)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s19
timeout 900 python -m pytest tests -m gpu -q -x -k "dt or dp or detect or batch" > gpurun_out/r04s19/pytest_dt.log 2>&1; echo "rc=$?" >> gpurun_out/r04s19/pytest_dt.log
tail -4 gpurun_out/r04s19/pytest_dt.log
for v in default ab_nested default ab_nested; do
  if [ $v = default ]; then unset PBD_LIBRARY; else export PBD_LIBRARY=$PWD/partsbaseddetector_amd/libpbd_hip_$v.so; fi
  timeout 300 python bench.py --legs batchseq,seq --no-prewarm --warmup 3 > gpurun_out/r04s19/bench_$v.json 2> gpurun_out/r04s19/bench_$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04s19/bench_$v.json').read().strip().splitlines()[-1])
print('$v', 'batched', d['stage_ms_per_frame_batched'], 'seq', d['stage_ms_sequential'])
PY
done
unset PBD_LIBRARY
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04s19/bench_driverflags.json 2> gpurun_out/r04s19/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04s19/bench_driverflags.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','value_incl_h2d','value_single_frame_calls')}, d['roofline']['frac'], d['roofline']['launch_ms'])
PY
