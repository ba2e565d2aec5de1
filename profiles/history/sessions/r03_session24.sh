#!/bin/bash
# session 24: brute-force DT core (PBD_DT_BF_MAXLEN, tuning build): parity of the whole GPU suite with it on, then timing
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03w
mkdir -p $OUT
cd $REPO
export PBD_LIBRARY=$REPO/partsbaseddetector_amd/libpbd_hip_tune.so
PBD_DT_BF_MAXLEN=256 timeout 900 python -m pytest tests -m gpu -q --maxfail=15 -k "not bench_lines" > $OUT/pytest_bf.log 2>&1; echo "pytest bf rc=$?" >> $OUT/summary.txt
tail -25 $OUT/pytest_bf.log | cut -c1-200 >> $OUT/summary.txt
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'single', d['value_single_frame_calls'], 'dp seq', d['stage_ms_sequential']['dp_min'], 'dp batched', (d.get('stage_ms_per_frame_batched') or {}).get('dp_min'), 'cands', d['config']['candidates_last_frame'])"; }
tp() { echo "$1: $(python bench.py --steps $3 --no-cpu-baseline $2 2>>$OUT/err.log | line)" >> $OUT/summary.txt; }
for b in 0 32 48 64 256; do
  PBD_DT_BF_MAXLEN=$b tp "bf maxlen $b S4 B3" "--inflight 4 --batch 3" 100
done
cat $OUT/summary.txt
