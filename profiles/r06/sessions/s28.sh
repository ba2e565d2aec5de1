# fused numerator / read-out (exact products): parity on the DT-heavy tests, then A/B against the previous build; issue-priority experiments (bank 1 / 3, DT 2)
mkdir -p gpurun_out/r06_s28
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dt2d or dp_min or detect_exact or person_full_size or fuzz or dp_pointers or trees or chains or mixtures or configs0 or 1080p" > gpurun_out/r06_s28/pytest_dt.log 2>&1
tail -2 gpurun_out/r06_s28/pytest_dt.log | cut -c1-200
bash profiles/r06/sessions/ab.sh r06_s28 3 libpbd_hip_base.so libpbd_hip.so libpbd_hip_bp3.so libpbd_hip_bp1.so libpbd_hip_dp2.so
