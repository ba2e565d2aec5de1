#!/bin/bash
# r04 session 11: k_root with a block table (no per-thread job search) and the fold instantiated for the model's mixture bound: full suite,
# then the evidence collection r04b (bench lines, batch-8-only and single-frame traces, HBM traffic, SQ counters, f64), and the two-rank gloo line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s11
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r04s11/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04s11/pytest.log
grep -E "passed|failed|rc=|^FAILED" gpurun_out/r04s11/pytest.log | tail -8
timeout 1500 bash profiles/collect_r04.sh r04b bench trace8 traceseq pmc8 sq f64 > gpurun_out/collect_r04b.log 2>&1
env -u RANK -u WORLD_SIZE -u LOCAL_RANK timeout 300 python bench.py --gpus 2 --backend gloo --steps 40 --warmup 5 --legs timed,h2d > gpurun_out/r04b/bench_gloo2.json 2> gpurun_out/r04b/bench_gloo2.err
python - <<'PY'
import json
for f in ('bench_n1.json','bench_n1_driverflags.json','bench_n1_b1.json','bench_n1_f64.json','bench_gloo2.json'):
    try:
        d=json.loads(open('gpurun_out/r04b/'+f).read().strip().splitlines()[-1])
        print(f, d['value'], d.get('value_incl_h2d'), d['roofline']['frac'] if d.get('roofline') else None, d.get('stage_ms_per_frame_batched'), d.get('stage_ms_sequential'))
    except Exception as e: print(f, 'failed', e)
PY
cat gpurun_out/r04b/batch_stages.txt | tail -5
