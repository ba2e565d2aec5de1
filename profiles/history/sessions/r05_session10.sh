#!/bin/bash
# r05 session 10: configs[4] table: direct VALU correlation vs fp32 MFMA vs split-product bank for 26 .. 312 filters (what PBD_CONV_AUTO should pick)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s10; mkdir -p $O
timeout 600 python profiles/conv_modes.py > $O/conv_modes.json 2> $O/conv_modes.err; echo "rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05s10/conv_modes.json').read().strip().splitlines()[-1])
for r in d['rows']: print(r)
PY
