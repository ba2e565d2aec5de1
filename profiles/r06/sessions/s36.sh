# fold loader: a child's plane pointers parked in a vector register (fetched with the part's own plane pointers / behind the previous child's loads) instead of a
# scalar round trip per child: parity (all fold / root / DT tests, float + double), per-phase trace of a fold launch, A/B against the previous build
mkdir -p gpurun_out/r06_s36
timeout 1300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_f64.py -x -q -m gpu -k "dt2d or dp_min or detect or person or fuzz or dp_pointers or trees or chains or mixtures or configs0 or 1080p or f64 or root or nms or group or batch or argmin" > gpurun_out/r06_s36/pytest_dt.log 2>&1
tail -2 gpurun_out/r06_s36/pytest_dt.log | cut -c1-200
python tests/tools_dt_trace.py 640 480 2 8 2>&1 | grep "batch of 8"
python tests/tools_dt_trace.py 640 480 2 2>&1 | grep -E "^(0|256|512|768) mean"
bash profiles/r06/sessions/ab.sh r06_s36 4 libpbd_hip_prev.so libpbd_hip.so
