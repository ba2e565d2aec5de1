timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dt2d or dp_min or detect_exact or person_full_size or fuzz_detect or random_models or group_batch" 2>&1 | tail -2
bash profiles/r06/sessions/ab.sh r06_s17 4 libpbd_hip_nolive.so libpbd_hip.so
