# the pointer-layout flag read from the group instead of the lane's map descriptor (no memory round trip in front of the loader): parity on the DT-heavy tests,
# A/B against the previous form (natmap); task prefetch 128 / 256 on top; natmap + prefetch 256 = session 30's best
mkdir -p gpurun_out/r06_s31
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_f64.py -x -q -m gpu -k "dt2d or dp_min or detect_exact or person_full_size or fuzz or dp_pointers or trees or chains or mixtures or configs0 or 1080p or f64" > gpurun_out/r06_s31/pytest_dt.log 2>&1
tail -2 gpurun_out/r06_s31/pytest_dt.log | cut -c1-200
bash profiles/r06/sessions/ab.sh r06_s31 3 libpbd_hip_natmap.so libpbd_hip.so libpbd_hip_tp128.so libpbd_hip_tp256.so libpbd_hip_natmap_tp256.so
