#!/bin/bash
# r04 session 22: the round's final state (commit 2bf9f1e kernels): full GPU suite, randomized stress against the oracle, evidence set r04c
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s22
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r04s22/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r04s22/pytest_gpu.log
tail -5 gpurun_out/r04s22/pytest_gpu.log
for s in 3 4; do timeout 300 python tests/tools_fuzz_detect.py 60 $s 2>&1 | tail -2 | tee -a gpurun_out/r04s22/fuzz.log; done
timeout 1500 bash profiles/collect_r04.sh r04c bench trace8 traceseq pmc8 sq trace1080 f64 > gpurun_out/r04s22/collect.log 2>&1
tail -3 gpurun_out/r04s22/collect.log
python - <<'PY'
import json
for f in ('bench_n1', 'bench_n1_driverflags', 'bench_n1_b1', 'bench_n1_f64'):
    try:
        d=json.loads(open('gpurun_out/r04c/%s.json' % f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d.get('roofline', {}).get('frac'), d.get('stage_ms_per_frame_batched'))
    except Exception as e:
        print(f, 'ERR', e)
PY
