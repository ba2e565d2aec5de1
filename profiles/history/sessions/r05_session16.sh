#!/bin/bash
# r05 session 16: the tree with PBD_CONV_SPLIT_F16 (opt-in) — whole GPU suite, smoke, the driver's bench command (with the split16 leg), fuzz on the binary16 bank, configs[4] table
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s16; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
timeout 500 python bench.py --steps 20 --warmup 5 > $O/bench_driverflags.json 2> $O/bench_driverflags.err; echo "bench rc=$?"
timeout 120 python tests/tools_fuzz_split.py 40 77 f16 > $O/fuzz_f16.log 2>&1; echo "rc=$?" >> $O/fuzz_f16.log; cat $O/fuzz_f16.log
timeout 400 python profiles/conv_modes.py > $O/conv_modes.json 2> $O/conv_modes.err; echo "rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05s16/bench_driverflags.json').read().strip().splitlines()[-1])
print('value', d['value'], 'pdf', d['pdf']['ms_per_frame_batched'], 'roof', d['roofline']['frac'], d['roofline']['launch_ms'], 'lat', d.get('latency_ms'), 'mfma32', d.get('value_fp32_mfma'))
print('split16', d.get('opt_in_split_f16'))
try:
    t = json.loads(open('gpurun_out/r05s16/conv_modes.json').read().strip().splitlines()[-1])
    for r in t['rows']: print(r['filters'], 'exact', r['exact_valu_ms'], 'mfma', r['mfma_f32_ms'], 'split', r['split_bf16x6_ms'], 'f16', r['split_f16x3_opt_in_ms'], r['split_f16x3_opt_in_tflops'], r['auto_picks'])
except Exception as e:
    print('conv_modes ERR', e)
PY
