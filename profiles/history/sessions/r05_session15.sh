#!/bin/bash
# r05 session 15: PBD_CONV_SPLIT_F16 with the filters two k-steps ahead (tuning variant 9) against the default; the restated range test
cd $GRAFT_REPO_ROOT
REPO=$GRAFT_REPO_ROOT
O=$REPO/gpurun_out/r05s15; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x -k "split_f16" > $O/pytest_f16.log 2>&1; echo "rc=$?" >> $O/pytest_f16.log; tail -4 $O/pytest_f16.log
for v in 0 9 0 9; do
  PBD_LIBRARY=$REPO/partsbaseddetector_amd/libpbd_hip_tune.so PBD_SPLIT_VARIANT=$v timeout 200 python bench.py --conv split16 --steps 100 --warmup 5 --legs timed,batchseq > $O/var$v.json 2> $O/var$v.err
  python - $v <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.loads(open(f'gpurun_out/r05s15/var{v}.json').read().strip().splitlines()[-1])
    print('variant', v, 'value', d['value'], 'pdf', d['pdf']['ms_per_frame_batched'], d['pdf']['TFLOP/s_batched'], 'dp', d['roofline']['launch_ms'])
except Exception as e:
    print('variant', v, 'ERR', e)
PY
done
PBD_LIBRARY=$REPO/partsbaseddetector_amd/libpbd_hip_tune.so PBD_SPLIT_VARIANT=9 timeout 300 python -m pytest tests -m gpu -q -x -k "split_products and f16x3" > $O/pytest_v9.log 2>&1; echo "rc=$?" >> $O/pytest_v9.log; tail -3 $O/pytest_v9.log
