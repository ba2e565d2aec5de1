#!/bin/bash
# session 34: five n-tiles per workgroup with 16-byte B loads (variant 23) vs the default (20)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03ae; mkdir -p $OUT; cd $REPO
export PBD_LIBRARY=$REPO/partsbaseddetector_amd/libpbd_hip_tune.so
PBD_MFMA_VARIANT=23 timeout 600 python -m pytest tests -m gpu -q -x -k "pdf_mfma_tol or detect_mfma or config5" > $OUT/pytest.log 2>&1; echo "pytest v23 rc=$?" > $OUT/summary.txt
tail -2 $OUT/pytest.log >> $OUT/summary.txt
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'pdf seq', d['stage_ms_sequential']['pdf'], 'pdf batched', (d.get('stage_ms_per_frame_batched') or {}).get('pdf'))"; }
tp() { echo "$1: $(python bench.py --steps 40 --no-cpu-baseline 2>>$OUT/err.log | line)" >> $OUT/summary.txt; }
PBD_MFMA_VARIANT=20 tp "variant 20"
PBD_MFMA_VARIANT=23 tp "variant 23 (NTW 5)"
PBD_MFMA_VARIANT=20 tp "variant 20"
PBD_MFMA_VARIANT=23 tp "variant 23 (NTW 5)"
cat $OUT/summary.txt
