#!/bin/bash
# r05 session 13: PBD_CONV_SPLIT_F16 (opt-in: two scaled binary16 parts per operand, three products) — parity, its stage times, the bench line with the split16 leg
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s13; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "split or auto_selects" > $O/pytest_split.log 2>&1; echo "rc=$?" >> $O/pytest_split.log; tail -5 $O/pytest_split.log
timeout 300 python bench.py --conv split16 --steps 100 --warmup 5 --legs timed,batchseq,seq > $O/bench_split16.json 2> $O/bench_split16.err; echo "rc=$?"
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?"
python - <<'PY'
import json
for n in ('bench_split16', 'bench_default'):
    try:
        d = json.loads(open(f'gpurun_out/r05s13/{n}.json').read().strip().splitlines()[-1])
        print(n, 'value', d['value'], 'conv', d['config'].get('conv'), 'pdf', d.get('pdf', {}).get('ms_per_frame_batched'), d.get('pdf', {}).get('TFLOP/s_batched'),
              'roof', d['roofline'].get('frac'), d['roofline'].get('launch_ms'), 'lat', d.get('latency_ms'), 'mfma32', d.get('value_fp32_mfma'), 'split16', d.get('opt_in_split_f16'))
    except Exception as e:
        print(n, 'ERR', e)
PY
tail -3 $O/*.err
