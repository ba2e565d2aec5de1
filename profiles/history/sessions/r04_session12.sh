#!/bin/bash
# r04 session 12: 1920x1080 evidence (kernel trace of single frames one at a time, reduced per kernel / chain) + bench lines at 1080p
cd $GRAFT_REPO_ROOT
timeout 900 bash profiles/collect_r04.sh r04b trace1080 > gpurun_out/collect_r04b_1080.log 2>&1
timeout 300 python bench.py --steps 20 --width 1920 --height 1080 --batch 2 --inflight 2 --legs timed,h2d,seq,batchseq > gpurun_out/r04b/bench_1080p.json 2> gpurun_out/r04b/bench_1080p.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04b/bench_1080p.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['roofline_single_frame']['frac'], d['stage_ms_per_frame_batched'], d['stage_ms_sequential'])
c=json.load(open('gpurun_out/r04b/seq1080_chains.json'))
for g in c['groups']: print(g['k_root_grid_threads'], g['chains'], round(g['sum_of_kernel_durations_ms'],4), {k:round(v['avg_us'],1) for k,v in g['per_kernel'].items()})
PY
