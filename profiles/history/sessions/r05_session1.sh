#!/bin/bash
# r05 session 1: the split-product filter bank's first run on the GPU.  (i) bf16-MFMA / VALU co-issue probe; (ii) the split bank's parity
# tests; (iii) its four kernel variants (tuning build: 2 / 4 wavefronts per workgroup x pinned / compiler schedule) timed through bench.py;
# (iv) the bench line with the driver's flags (AUTO -> split) and with the fp32 MFMA bank; (v) the whole GPU suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s1; mkdir -p $O
timeout 120 tests/tools/mfma_valu_overlap_probe > $O/overlap_probe.log 2>&1; echo "rc=$?" >> $O/overlap_probe.log
cat $O/overlap_probe.log
timeout 900 python -m pytest tests -m gpu -q -x -k "split or auto_selects or pdf_" > $O/pytest_split.log 2>&1; echo "rc=$?" >> $O/pytest_split.log
tail -15 $O/pytest_split.log
for v in 0 1 2 3; do
  PBD_LIBRARY=$PWD/partsbaseddetector_amd/libpbd_hip_tune.so PBD_SPLIT_VARIANT=$v timeout 300 python bench.py --steps 60 --conv split --legs timed,seq,batchseq 2> $O/var$v.err > $O/var$v.json
  python - $O/var$v.json $v <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('split variant', sys.argv[2], 'value', d['value'], 'batched', d['stage_ms_per_frame_batched'], 'seq', d['stage_ms_sequential'])
except Exception as e: print('variant', sys.argv[2], 'failed', e)
PY
done
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_driverflags.json 2> $O/bench.err
timeout 300 python bench.py --steps 200 --legs timed,h2d,mfma32,batchseq > $O/bench_200.json 2>> $O/bench.err
timeout 300 python bench.py --steps 100 --conv mfma --legs timed,seq,batchseq > $O/bench_mfma.json 2>> $O/bench.err
python - <<'PY'
import json
for f in ('bench_driverflags','bench_200','bench_mfma'):
    try:
        d=json.loads(open('gpurun_out/r05s1/%s.json'%f).read().strip().splitlines()[-1])
        print(f, d['value'], 'mfma32', d.get('value_fp32_mfma'), 'roof', d['roofline']['frac'], d['stage_ms_per_frame_batched'], d['stage_ms_sequential'], d['pdf'])
    except Exception as e: print(f, 'failed', e)
PY
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log
tail -25 $O/pytest_all.log
