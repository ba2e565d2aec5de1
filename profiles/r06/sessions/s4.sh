timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dt2d or dp_min or detect_exact or person_full_size or fuzz_detect or random_models" 2>&1 | tail -2
bash profiles/r06/sessions/ab.sh r06_s4 3 libpbd_hip_base.so libpbd_hip.so
for l in 0 1 2; do python tests/tools_dt_trace.py 640 480 $l 16 2>&1 | grep "batch of"; done
python tests/tools_dt_trace.py 640 480 0 2>&1 | sed -n 3,3p
