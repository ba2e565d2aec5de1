#!/bin/bash
# r05 session 9: the final tree: whole GPU suite, smoke, the bench line with the driver's flags
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s9; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log; tail -5 $O/pytest_all.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_driverflags.json 2> $O/bench_driverflags.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05s9/bench_driverflags.json').read().strip().splitlines()[-1])
print('driverflags value', d['value'], 'ms/step', d['ms_per_step'], 'mfma32', d.get('value_fp32_mfma'), 'roof', d['roofline']['frac'], d['stage_ms_per_frame_batched'], 'cpu', d['cpu_baseline']['value'])
PY
