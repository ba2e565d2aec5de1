#!/bin/bash
# r04 session 18: fold loader without the K == 1 branch / the zero-trip children loop (all child loads in one memory round trip, no wait for
# the Ik stores in front of the LDS stores), map descriptor fetched in front of the loader; with session 16's read-out walk.  DT parity tests,
# then A/B of the dp_min stage: default (new) / ab_old (the kernels of commit dde734c, before session 16)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s18
timeout 900 python -m pytest tests -m gpu -q -x -k "dt or dp or detect or batch" > gpurun_out/r04s18/pytest_dt.log 2>&1; echo "rc=$?" >> gpurun_out/r04s18/pytest_dt.log
tail -4 gpurun_out/r04s18/pytest_dt.log
for v in default ab_old default ab_old; do
  if [ $v = default ]; then unset PBD_LIBRARY; else export PBD_LIBRARY=$PWD/partsbaseddetector_amd/libpbd_hip_$v.so; fi
  timeout 300 python bench.py --legs batchseq,seq --no-prewarm --warmup 3 > gpurun_out/r04s18/bench_$v.json 2> gpurun_out/r04s18/bench_$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04s18/bench_$v.json').read().strip().splitlines()[-1])
print('$v', 'batched', d['stage_ms_per_frame_batched'], 'seq', d['stage_ms_sequential'])
PY
done
unset PBD_LIBRARY
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04s18/bench_driverflags.json 2> gpurun_out/r04s18/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04s18/bench_driverflags.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','value_incl_h2d','value_single_frame_calls')}, d['roofline']['frac'], d['roofline']['launch_ms'])
PY
