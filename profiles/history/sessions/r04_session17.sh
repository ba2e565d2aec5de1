#!/bin/bash
# r04 session 17: (1) what bounds the read-out if not vector issue?  timing-only variant ab_tr: x-pass pointers written transposed
# (coalesced across the lanes of a sub-range) instead of natural (one 2-byte run per lane): dp_min A/B (results of ab_tr are NOT valid);
# (2) per-phase block times on a loaded chip: probe build, batch of 8, launches 0 (x pass), 1 (y pass), 2 (fold x pass)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s17
for v in default ab_tr default ab_tr; do
  if [ $v = default ]; then unset PBD_LIBRARY; else export PBD_LIBRARY=$PWD/partsbaseddetector_amd/libpbd_hip_$v.so; fi
  timeout 300 python bench.py --legs batchseq,seq --no-prewarm --warmup 3 > gpurun_out/r04s17/bench_$v.json 2> gpurun_out/r04s17/bench_$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04s17/bench_$v.json').read().strip().splitlines()[-1])
print('$v', 'batched', d['stage_ms_per_frame_batched'], 'seq', d['stage_ms_sequential'])
PY
done
unset PBD_LIBRARY
for l in 0 1 2; do timeout 200 python tests/tools_dt_trace.py 640 480 $l 8 > gpurun_out/r04s17/trace8_launch$l.txt 2> gpurun_out/r04s17/trace8_launch$l.err; grep "batch of" gpurun_out/r04s17/trace8_launch$l.txt; done
head -4 gpurun_out/r04s17/trace8_launch0.txt
