# per-lane z table merged (ZSAVE shares ZLO's slots: 4 B per lane less LDS, 1 040 -> 1 016 fold blocks per single frame launch): parity on the DT-heavy tests, A/B against the
# fused-only build; issue priority of the DT wavefronts 1 / 2 / 3 and DT + HOG at 3 (over the merged-table build)
mkdir -p gpurun_out/r06_s29
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_f64.py -x -q -m gpu -k "dt2d or dp_min or detect_exact or person_full_size or fuzz or dp_pointers or trees or chains or mixtures or configs0 or 1080p or f64" > gpurun_out/r06_s29/pytest_dt.log 2>&1
tail -2 gpurun_out/r06_s29/pytest_dt.log | cut -c1-200
bash profiles/r06/sessions/ab.sh r06_s29 3 libpbd_hip_fz.so libpbd_hip.so libpbd_hip_dp1.so libpbd_hip_dp2.so libpbd_hip_dp3.so libpbd_hip_dh3.so
