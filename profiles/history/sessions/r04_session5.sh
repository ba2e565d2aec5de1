#!/bin/bash
# r04 session 5: DT fold_children / setup-division diet on top of the new k_hog: full suite, bench line, batch-8 trace + SQ counters
# features): HOG / pyramid parity first, then the full suite, bench line, batch-8 trace + SQ counters
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s5
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r04s5/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04s5/pytest.log
grep -E "passed|failed|rc=|^FAILED" gpurun_out/r04s5/pytest.log | tail -12
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04s5/bench_driverflags.json 2> gpurun_out/r04s5/bench.err
timeout 600 bash profiles/collect_r04.sh r04s5 trace8 sq > gpurun_out/r04s5/collect.log 2>&1
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04s5/bench_driverflags.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','value_incl_h2d','value_single_frame_calls')}, d['roofline']['frac'], d['roofline']['launch_ms'], d['stage_ms_per_frame_batched'], d['stage_ms_sequential'])
PY
grep "k_hog\|k_resize\|k_pyrdown" gpurun_out/r04s5/batch8_kernel_stats.csv
