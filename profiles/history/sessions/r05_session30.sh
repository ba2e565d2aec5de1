#!/bin/bash
# r05 session 30: the double instantiation and 1920x1080 at the new default of 16 frames per batch (do they run, what do they give)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s30; mkdir -p $O
timeout 200 python bench.py --dtype f64 --steps 30 --warmup 3 --legs timed,batchseq > $O/bench_f64.json 2> $O/bench_f64.err; echo "f64 rc=$?"
timeout 250 python bench.py --width 1920 --height 1080 --steps 10 --warmup 2 --legs timed,batchseq > $O/bench_1080p.json 2> $O/bench_1080p.err; echo "1080p rc=$?"
timeout 200 python bench.py --width 1920 --height 1080 --batch 8 --steps 20 --warmup 2 --legs timed > $O/bench_1080p_b8.json 2> $O/bench_1080p_b8.err; echo "1080p b8 rc=$?"
python - <<'PY'
import json
for n in ('bench_f64', 'bench_1080p', 'bench_1080p_b8'):
    try:
        d = json.loads(open(f'gpurun_out/r05s30/{n}.json').read().strip().splitlines()[-1])
        print(n, 'value', d['value'], 'batch', d['config'].get('frames_per_batch'), d.get('stage_ms_per_frame_batched'), 'roof', d.get('roofline', {}).get('frac'))
    except Exception as e:
        print(n, 'ERR', e)
PY
tail -2 $O/*.err
