#!/bin/bash
# r04 session 8: evidence collection r04a (bench lines, batch-8-only and single-frame kernel traces, HBM traffic and SQ counters of the batch chains)
cd $GRAFT_REPO_ROOT
timeout 1500 bash profiles/collect_r04.sh r04a bench trace8 traceseq pmc8 sq > gpurun_out/collect_r04a.log 2>&1
tail -3 gpurun_out/collect_r04a.log
python - <<'PY'
import json
for f in ('bench_n1.json','bench_n1_driverflags.json','bench_n1_b1.json'):
    d=json.loads(open('gpurun_out/r04a/'+f).read().strip().splitlines()[-1])
    print(f, d['value'], d.get('value_incl_h2d'), d['roofline']['frac'] if d.get('roofline') else None, d.get('stage_ms_per_frame_batched'))
PY
cat gpurun_out/r04a/batch_stages.txt | tail -8
