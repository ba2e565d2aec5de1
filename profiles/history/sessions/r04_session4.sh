#!/bin/bash
# r04 session 4: new k_hog (4-byte staging loads, tabulated orientation snap, compile-time histogram walk, one thread per cell for the
# features): HOG / pyramid parity first, then the full suite, bench line, batch-8 trace + SQ counters
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s4
timeout 600 python -m pytest tests -m gpu -q -x -k "hog or pyramid or features or smoke or detect_small" > gpurun_out/r04s4/pytest_hog.log 2>&1; echo "rc=$?" >> gpurun_out/r04s4/pytest_hog.log
tail -4 gpurun_out/r04s4/pytest_hog.log
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r04s4/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04s4/pytest.log
grep -E "passed|failed|rc=|^FAILED" gpurun_out/r04s4/pytest.log | tail -12
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04s4/bench_driverflags.json 2> gpurun_out/r04s4/bench.err
timeout 600 bash profiles/collect_r04.sh r04s4 trace8 sq > gpurun_out/r04s4/collect.log 2>&1
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04s4/bench_driverflags.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','value_incl_h2d','value_single_frame_calls')}, d['roofline']['frac'], d['roofline']['launch_ms'], d['stage_ms_per_frame_batched'], d['stage_ms_sequential'])
PY
grep "k_hog\|k_resize\|k_pyrdown" gpurun_out/r04s4/batch8_kernel_stats.csv
