#!/bin/bash
# session 33: 16-byte B loads in the double filter bank + the tuning-variant parity test
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03ad; mkdir -p $OUT; cd $REPO
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 -k "f64 or double or float64 or tuning_variants or mfma or pdf" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" > $OUT/summary.txt
tail -3 $OUT/pytest.log | cut -c1-200 >> $OUT/summary.txt
python bench.py --dtype f64 --steps 20 --no-cpu-baseline 2>>$OUT/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('f64 default', d['value'], d['value_single_frame_calls'], d['stage_ms_sequential'], d['pdf'])" >> $OUT/summary.txt
python bench.py --dtype f64 --steps 60 --batch 1 --inflight 4 --no-cpu-baseline 2>>$OUT/err.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('f64 single frames x4', d['value'])" >> $OUT/summary.txt
cat $OUT/summary.txt
