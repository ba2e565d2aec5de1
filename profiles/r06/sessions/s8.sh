timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wide_depths_graph or dt2d or dp_min or detect_exact or person_full_size or fuzz_detect or random_models or compact or 1080p_levels or three_kernel or foreign or batch_equals" 2>&1 | tail -3
bash profiles/r06/sessions/ab.sh r06_s8 3 libpbd_hip_v1.so libpbd_hip.so
for l in 0 1 2; do python tests/tools_dt_trace.py 640 480 $l 16 2>&1 | grep "batch of"; done
