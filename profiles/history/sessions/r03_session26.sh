#!/bin/bash
# session 26: brute-force DT core with the sign handled: redo counts / phases, parity (dp + dt subset), timing sweep
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03y
mkdir -p $OUT
cd $REPO
PBD_DT_BF_MAXLEN=256 python tests/tools_dt_trace.py 640 480 1 > $OUT/trace_bf_l1.txt 2>$OUT/err.log
head -3 $OUT/trace_bf_l1.txt | cut -c1-200 >> $OUT/summary.txt
grep -v "^launch" $OUT/trace_bf_l1.txt | sed -n 2,26p >> $OUT/summary.txt
export PBD_LIBRARY=$REPO/partsbaseddetector_amd/libpbd_hip_tune.so
PBD_DT_BF_MAXLEN=256 timeout 900 python -m pytest tests -m gpu -q --maxfail=15 -k "dt or dp or detect or fold or batch" > $OUT/pytest_bf.log 2>&1; echo "pytest bf rc=$?" >> $OUT/summary.txt
tail -4 $OUT/pytest_bf.log | cut -c1-200 >> $OUT/summary.txt
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'single', d['value_single_frame_calls'], 'dp seq', d['stage_ms_sequential']['dp_min'], 'dp batched', (d.get('stage_ms_per_frame_batched') or {}).get('dp_min'), 'cands', d['config']['candidates_last_frame'])"; }
tp() { echo "$1: $(python bench.py --steps $3 --no-cpu-baseline $2 2>>$OUT/err.log | line)" >> $OUT/summary.txt; }
for b in 0 32 48 64 256; do
  PBD_DT_BF_MAXLEN=$b tp "bf maxlen $b S4 B3" "--inflight 4 --batch 3" 100
done
cat $OUT/summary.txt
