#!/bin/bash
# r04 session 30: the filter bank as committed (epilogue stores with a uniform plane base + 32-bit cell offset, tile-local divisions as
# multiplies; the staging as before): parity tests of everything that touches the bank, one driver-flag bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s30
timeout 900 python -m pytest tests -m gpu -q -x -k "pdf or filter or 7x7 or conv or mfma or config5 or timed_configuration or tuning or benched_unit or f64" > gpurun_out/r04s30/pytest_conv.log 2>&1; echo "rc=$?" >> gpurun_out/r04s30/pytest_conv.log
tail -3 gpurun_out/r04s30/pytest_conv.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04s30/bench_driverflags.json 2> gpurun_out/r04s30/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04s30/bench_driverflags.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','value_incl_h2d','value_single_frame_calls')}, d['roofline']['frac'], d['stage_ms_per_frame_batched'])
PY
