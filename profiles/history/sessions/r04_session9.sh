#!/bin/bash
# r04 session 9: k_hog gradient phase in unpredicated batches of four pixels (table look-ups in flight together): parity, then the S x B sweep
# (handles in flight x frames per batch) with the round's faster kernels, the double instantiation and 1920x1080
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s9
timeout 900 python -m pytest tests -m gpu -q -x -k "hog or pyramid or features or detect or smoke" > gpurun_out/r04s9/pytest_hog.log 2>&1; echo "rc=$?" >> gpurun_out/r04s9/pytest_hog.log
tail -3 gpurun_out/r04s9/pytest_hog.log
for sb in "3 8" "4 8" "2 8" "3 12" "4 6" "3 16" "2 16"; do set -- $sb
  timeout 200 python bench.py --steps 60 --inflight $1 --batch $2 --legs timed 2> gpurun_out/r04s9/sweep.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight $1 batch $2:', d['value'], 'frames/s')" | tee -a gpurun_out/r04s9/sweep_sb.txt
done
timeout 300 python bench.py --steps 20 --warmup 5 --legs timed,h2d,seq,batchseq > gpurun_out/r04s9/bench_driverflags.json 2> gpurun_out/r04s9/bench.err
timeout 300 python bench.py --steps 50 --dtype f64 --legs timed,seq,batchseq > gpurun_out/r04s9/bench_f64.json 2>> gpurun_out/r04s9/bench.err
timeout 300 python bench.py --steps 20 --width 1920 --height 1080 --batch 2 --inflight 2 --legs timed,seq,batchseq > gpurun_out/r04s9/bench_1080p.json 2>> gpurun_out/r04s9/bench.err
python - <<'PY'
import json
for f in ('bench_driverflags','bench_f64','bench_1080p'):
    try:
        d=json.loads(open('gpurun_out/r04s9/%s.json'%f).read().strip().splitlines()[-1])
        print(f, d['value'], d['roofline']['frac'], d['stage_ms_per_frame_batched'], d['stage_ms_sequential'])
    except Exception as e: print(f, 'failed', e)
PY
