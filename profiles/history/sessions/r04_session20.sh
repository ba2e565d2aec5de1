#!/bin/bash
# r04 session 20: wave priorities inside k_dt_pass (s_setprio), timing A/B: default / ab_prio1 (scans + stitches, the issue-bound phases, at
# priority 1) / ab_prio2 (loader + read-out, the memory phases, at priority 1): dp_min stage alone and the whole pipeline (3 handles in flight)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s20
for v in default ab_prio1 ab_prio2 default ab_prio1 ab_prio2; do
  if [ $v = default ]; then unset PBD_LIBRARY; else export PBD_LIBRARY=$PWD/partsbaseddetector_amd/libpbd_hip_$v.so; fi
  timeout 300 python bench.py --steps 150 --legs timed,batchseq,seq --no-prewarm --warmup 5 > gpurun_out/r04s20/bench_$v.json 2> gpurun_out/r04s20/bench_$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04s20/bench_$v.json').read().strip().splitlines()[-1])
print('$v', d['value'], 'batched dp_min', d['stage_ms_per_frame_batched']['dp_min'], 'pdf', d['stage_ms_per_frame_batched']['pdf'], 'seq dp_min', d['stage_ms_sequential']['dp_min'])
PY
done
