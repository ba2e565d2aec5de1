timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dt2d or dp_min or detect_exact or person_full_size or fuzz_detect or random_models" 2>&1 | tail -2
bash profiles/r06/sessions/ab.sh r06_s5 3 libpbd_hip_v1.so libpbd_hip_mplds.so
