#!/bin/bash
# r04 session 29: filter bank with fewer non-MFMA vector instructions (staging plan computed once, epilogue stores with a uniform plane base +
# 32-bit cell offset, tile-local divisions as multiplies): pdf / detector parity tests, then A/B default (new) / ab_old (commit before)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s29
timeout 900 python -m pytest tests -m gpu -q -x -k "pdf or filter or 7x7 or conv or mfma or config5 or timed_configuration or tuning or benched_unit" > gpurun_out/r04s29/pytest_conv.log 2>&1; echo "rc=$?" >> gpurun_out/r04s29/pytest_conv.log
tail -4 gpurun_out/r04s29/pytest_conv.log
for v in default ab_old default ab_old; do
  if [ $v = default ]; then unset PBD_LIBRARY; else export PBD_LIBRARY=$PWD/partsbaseddetector_amd/libpbd_hip_$v.so; fi
  timeout 300 python bench.py --steps 200 --legs timed,batchseq,seq --warmup 5 --no-cpu-baseline > gpurun_out/r04s29/bench_$v.json 2> gpurun_out/r04s29/bench_$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04s29/bench_$v.json').read().strip().splitlines()[-1])
print('$v', d['value'], 'batched pdf', d['stage_ms_per_frame_batched']['pdf'], 'dp_min', d['stage_ms_per_frame_batched']['dp_min'], 'seq pdf', d['stage_ms_sequential']['pdf'])
PY
done
