#!/bin/bash
# r05 session 11: balanced n-tile groups for banks of more than 160 filters (208 filters: 4 + 3 instead of 5 + 2): configs[4] table again, parity, fuzz
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s11; mkdir -p $O
timeout 600 python profiles/conv_modes.py > $O/conv_modes.json 2> $O/conv_modes.err; echo "rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05s11/conv_modes.json').read().strip().splitlines()[-1])
for r in d['rows']: print(r['filters'], 'exact', r['exact_valu_ms'], 'mfma', r['mfma_f32_ms'], 'split', r['split_bf16x6_ms'], r['split_bf16x6_tflops'], r['auto_picks'])
PY
timeout 900 python -m pytest tests -m gpu -q -x -k "config5 or split or benched" > $O/pytest_sub.log 2>&1; echo "rc=$?" >> $O/pytest_sub.log; tail -3 $O/pytest_sub.log
timeout 200 python tests/tools_fuzz_split.py 45 31 > $O/fuzz_split.log 2>&1; echo "rc=$?" >> $O/fuzz_split.log; cat $O/fuzz_split.log
