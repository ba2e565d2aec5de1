#!/bin/bash
# r04 session 21: persistent k_dt_pass blocks for plain launches larger than the chip, fetching the next task's lines into registers under the
# current task's stitches.  DT / detector parity tests (batches exercise the persistent path), then A/B in the tuning build:
# PBD_DT_PERSIST=1 / 0, dp_min stage + whole pipeline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s21
timeout 900 python -m pytest tests -m gpu -q -x -k "dt or dp or detect or batch" > gpurun_out/r04s21/pytest_dt.log 2>&1; echo "rc=$?" >> gpurun_out/r04s21/pytest_dt.log
tail -4 gpurun_out/r04s21/pytest_dt.log
export PBD_LIBRARY=$PWD/partsbaseddetector_amd/libpbd_hip_tune.so
for v in 1 0 1 0; do
  PBD_DT_PERSIST=$v timeout 300 python bench.py --steps 150 --legs timed,batchseq,seq --warmup 5 > gpurun_out/r04s21/bench_persist$v.json 2> gpurun_out/r04s21/bench_persist$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04s21/bench_persist$v.json').read().strip().splitlines()[-1])
print('persist=$v', d['value'], 'batched dp_min', d['stage_ms_per_frame_batched']['dp_min'], 'pdf', d['stage_ms_per_frame_batched']['pdf'], 'seq dp_min', d['stage_ms_sequential']['dp_min'])
PY
done
