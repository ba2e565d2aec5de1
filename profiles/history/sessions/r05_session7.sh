#!/bin/bash
# r05 session 7: the round's evidence with the final kernels: bench lines (300 steps, driver flags, single frames in flight, T = double, 1920x1080),
# kernel traces of batch-of-8 and single-frame chains, HBM traffic passes, SQ counters (issue, lane utilisation, matrix pipe); the whole GPU suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s7; mkdir -p $O
bash profiles/collect_r05.sh r05f bench trace8 traceseq pmc8 sq sq2 sq3 f64 > $O/collect.log 2>&1; tail -3 $O/collect.log
timeout 300 python bench.py --width 1920 --height 1080 --steps 10 --legs timed,seq,batchseq > gpurun_out/r05f/bench_1080p.json 2> gpurun_out/r05f/bench_1080p.err
python - <<'PY'
import json
for f in ('bench_n1','bench_n1_driverflags','bench_n1_b1','bench_n1_f64','bench_1080p'):
    try:
        d=json.loads(open('gpurun_out/r05f/%s.json'%f).read().strip().splitlines()[-1])
        print(f, d['value'], 'ms/step', d['ms_per_step'], 'mfma32', d.get('value_fp32_mfma'), 'roof', d['roofline']['frac'], d.get('stage_ms_per_frame_batched'), d.get('stage_ms_sequential'), (d.get('sequential') or {}).get('latency_ms'))
    except Exception as e: print(f, 'failed', e)
PY
timeout 1700 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
tail -6 $O/pytest_all.log
