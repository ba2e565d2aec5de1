#!/bin/bash
# r05 session 22: the tree with pbd_detect_image and pbd_tune_plan — whole GPU suite, smoke, the driver's bench command
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s22; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; grep -E "passed|failed" $O/pytest_gpu.log | tail -3
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
timeout 500 python bench.py --steps 20 --warmup 5 > $O/bench_driverflags.json 2> $O/bench_driverflags.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05s22/bench_driverflags.json').read().strip().splitlines()[-1])
print('value', d['value'], 'pdf', d['pdf']['ms_per_frame_batched'], 'roof', d['roofline']['frac'], d['roofline']['launch_ms'], 'lat', d['sequential'].get('latency_ms'), 'mfma32', d.get('value_fp32_mfma'), 'split16', (d.get('opt_in_split_f16') or {}).get('value'))
print(d['stage_ms_per_frame_batched'])
PY
