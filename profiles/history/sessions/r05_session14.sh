#!/bin/bash
# r05 session 14: PBD_CONV_SPLIT_F16 — the restated range test, tuning variants (hipcc's schedule, n-tile groups of 4 / 3), matrix-pipe counters
cd $GRAFT_REPO_ROOT
REPO=$GRAFT_REPO_ROOT
O=$REPO/gpurun_out/r05s14; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x -k "split_f16" > $O/pytest_f16.log 2>&1; echo "rc=$?" >> $O/pytest_f16.log; tail -4 $O/pytest_f16.log
for v in 0 4 7 8 0; do
  PBD_LIBRARY=$REPO/partsbaseddetector_amd/libpbd_hip_tune.so PBD_SPLIT_VARIANT=$v timeout 200 python bench.py --conv split16 --steps 100 --warmup 5 --legs timed,batchseq > $O/var$v.json 2> $O/var$v.err
  python - $v <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.loads(open(f'gpurun_out/r05s14/var{v}.json').read().strip().splitlines()[-1])
    print('variant', v, 'value', d['value'], 'pdf', d['pdf']['ms_per_frame_batched'], d['pdf']['TFLOP/s_batched'], 'dp', d['roofline']['launch_ms'])
except Exception as e:
    print('variant', v, 'ERR', e)
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES \
  --kernel-trace --output-format csv -d "$O/sq16" -o run -- python $REPO/bench.py --conv split16 --legs batchseq --graph 0 --inflight 1 --no-prewarm --warmup 2 > "$O/sq16.log" 2>&1
find "$O/sq16" -name "*kernel_trace.csv" -delete; find "$O/sq16" -name "*agent_info.csv" -delete
python - <<'PY'
import csv, glob, collections
f = glob.glob('/root/repo/gpurun_out/r05s14/sq16/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'].split('(')[0][:40]
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k in acc:
    if 'conv_split' in k or 'dt_pass' in k:
        print(k, {c: round(v / max(1, cnt[(k, c)])) for c, v in acc[k].items()})
PY
du -sh $O
