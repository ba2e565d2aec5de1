#!/bin/bash
# r04 session 28: final state of the round (256-lane DT blocks at 640x480): full GPU suite, evidence set r04d (bench lines, batch-8 and
# single-frame kernel traces, HBM traffic passes, SQ counters)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s28
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r04s28/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r04s28/pytest_gpu.log
tail -3 gpurun_out/r04s28/pytest_gpu.log
timeout 1500 bash profiles/collect_r04.sh r04d bench trace8 traceseq pmc8 sq > gpurun_out/r04s28/collect.log 2>&1
tail -3 gpurun_out/r04s28/collect.log
python - <<'PY'
import json
for f in ('bench_n1', 'bench_n1_driverflags', 'bench_n1_b1'):
    try:
        d=json.loads(open('gpurun_out/r04d/%s.json' % f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d.get('roofline', {}).get('frac'), d.get('stage_ms_per_frame_batched'))
    except Exception as e:
        print(f, 'ERR', e)
PY
