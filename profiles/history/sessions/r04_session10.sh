#!/bin/bash
# r04 session 10: fold loader with scalar-base + 32-bit-offset loads / stores (offsets kept out of the loop-invariant 64-bit form), read-out
# loop counted on the shifted position: DT / DP / end-to-end parity, bench line, batch-8 trace + SQ counters
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s10
timeout 900 python -m pytest tests -m gpu -q -x -k "dt or dp or detect or fold or person or config or batch or compact" > gpurun_out/r04s10/pytest_dt.log 2>&1; echo "rc=$?" >> gpurun_out/r04s10/pytest_dt.log
tail -3 gpurun_out/r04s10/pytest_dt.log
timeout 300 python bench.py --steps 100 --warmup 5 > gpurun_out/r04s10/bench.json 2> gpurun_out/r04s10/bench.err
timeout 600 bash profiles/collect_r04.sh r04s10 trace8 sq > gpurun_out/r04s10/collect.log 2>&1
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04s10/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','value_incl_h2d','value_single_frame_calls')}, d['roofline']['frac'], d['roofline']['launch_ms'], d['stage_ms_per_frame_batched'], d['stage_ms_sequential'])
c=json.load(open('gpurun_out/r04s10/batch8_chains.json'))
for g in c['groups']: print(g['k_root_grid_threads'], g['chains'], round(g['sum_of_kernel_durations_ms'],4), {k:round(v['avg_us'],1) for k,v in g['per_kernel'].items()})
PY
