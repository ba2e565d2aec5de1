set -u
O=gpurun_out/r06_s2; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dt2d or dp_min or detect_exact or person_full_size or fuzz_detect or random_models or tune_plan or detect_dt2d" > $O/pytest_dt.log 2>&1
tail -3 $O/pytest_dt.log
python bench.py --steps 20 --warmup 5 > $O/bench_driverflags.json 2> $O/bench.err
for l in 0 1 2; do
  python tests/tools_dt_trace.py 640 480 $l > $O/trace_single_l$l.txt 2>&1
  python tests/tools_dt_trace.py 640 480 $l 16 > $O/trace_b16_l$l.txt 2>&1
done
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_s2/bench_driverflags.json'))
print(d['value'], d['roofline']['frac'], d['roofline']['launch_ms'], d['stage_ms_per_frame_batched'], d['stage_ms_sequential'], d.get('value_single_frame_calls'))
PY
grep "batch of" $O/trace_b16_l*.txt
