#!/bin/bash
# r05 session 25: bench.py's default batch 8 -> 16 (sessions 23 / 24): the evidence set again (tag r05g: bench lines, batch-chain kernel trace, HBM traffic passes),
# and the benched unit against the oracle at the new batch size
cd $GRAFT_REPO_ROOT
bash profiles/collect_r05.sh r05g bench trace8 pmc8 > gpurun_out/r05g_collect.log 2>&1; echo "collect rc=$?"
O=gpurun_out/r05s25; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -s -k "benched_unit or bench_line or bench_py" > $O/pytest_bench.log 2>&1; echo "rc=$?" >> $O/pytest_bench.log; grep -E "default bank vs oracle|passed|failed|rror" $O/pytest_bench.log | tail -6
python - <<'PY'
import json
for n in ('bench_n1', 'bench_n1_driverflags'):
    d = json.loads(open(f'gpurun_out/r05g/{n}.json').read().strip().splitlines()[-1])
    print(n, 'value', d['value'], 'step', d['ms_per_step'], 'pdf', d['pdf']['ms_per_frame_batched'], 'roof', d['roofline']['frac'], d['roofline']['launch_ms'], 'traffic', d['roofline']['traffic'],
          'lat', d['sequential'].get('latency_ms', {}).get('median'), 'mfma32', d.get('value_fp32_mfma'), 'split16', (d.get('opt_in_split_f16') or {}).get('value'), 'h2d', d.get('value_incl_h2d'), 'single', d.get('value_single_frame_calls'))
    print('   ', d['stage_ms_per_frame_batched'])
PY
cat gpurun_out/r05g/batch_stages.txt | tail -8
