#!/bin/bash
# r04 session 1: full GPU suite on the round's first changes (ADVICE fixes, new bench legs, batch-8 oracle test), driver-flag bench line,
# batch-8-only kernel trace + one-pass SQ counters (the "before" of the round's DT work)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04s1/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04s1/pytest.log
tail -5 gpurun_out/r04s1/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04s1/bench_driverflags.json 2> gpurun_out/r04s1/bench.err
timeout 600 bash profiles/collect_r04.sh r04s1 trace8 sq > gpurun_out/r04s1/collect.log 2>&1
cat gpurun_out/r04s1/batch8_reduce.log | head -60
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04s1/bench_driverflags.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','value_incl_h2d','value_single_frame_calls')}, d['roofline']['frac'], d['roofline']['launch_ms'], d['stage_ms_per_frame_batched'], d['stage_ms_sequential'])
PY
