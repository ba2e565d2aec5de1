timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wide_depths_graph or dp_min or detect_exact or person_full_size or fuzz_detect or random_models or compact or batch_equals or face_like" 2>&1 | tail -3
bash profiles/r06/sessions/ab.sh r06_s9 3 libpbd_hip_v2.so libpbd_hip.so
python tests/tools_dt_trace.py 640 480 2 16 2>&1 | grep "batch of"
