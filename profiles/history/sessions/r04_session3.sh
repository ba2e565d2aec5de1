#!/bin/bash
# r04 session 3: new dt_core (lean scan / flat stitch / min-accumulated suspect test), loaders with one index computation per element:
# full GPU suite, driver-flag bench line, batch-8 trace + SQ counters
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s3
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/r04s3/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04s3/pytest.log
grep -E "MFMA bank vs oracle|passed|failed|rc=|^FAILED" gpurun_out/r04s3/pytest.log | tail -12
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04s3/bench_driverflags.json 2> gpurun_out/r04s3/bench.err
timeout 600 bash profiles/collect_r04.sh r04s3 trace8 sq > gpurun_out/r04s3/collect.log 2>&1
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04s3/bench_driverflags.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','value_incl_h2d','value_single_frame_calls')}, d['roofline']['frac'], d['roofline']['launch_ms'], d['stage_ms_per_frame_batched'], d['stage_ms_sequential'])
c=json.load(open('gpurun_out/r04s3/batch8_chains.json'))
for g in c['groups']: print(g['k_root_grid_threads'], g['chains'], g['sum_of_kernel_durations_ms'], {k:round(v['avg_us'],1) for k,v in g['per_kernel'].items()})
PY
