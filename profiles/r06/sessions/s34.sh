# dt_cover: four segments per LDS round trip (the walk in front of a lane's first output): parity on the DT-heavy tests, per-phase trace, A/B against the previous build
mkdir -p gpurun_out/r06_s34
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_f64.py -x -q -m gpu -k "dt2d or dp_min or detect_exact or person_full_size or fuzz or dp_pointers or trees or chains or mixtures or configs0 or 1080p or f64" > gpurun_out/r06_s34/pytest_dt.log 2>&1
tail -2 gpurun_out/r06_s34/pytest_dt.log | cut -c1-200
for l in 0 1 2; do python tests/tools_dt_trace.py 640 480 $l 8 2>&1 | grep "batch of 8"; done
for l in 1 2; do python tests/tools_dt_trace.py 640 480 $l 2>&1 | grep -E "^(0|256|512|768) mean"; done
bash profiles/r06/sessions/ab.sh r06_s34 4 libpbd_hip_prev.so libpbd_hip.so
