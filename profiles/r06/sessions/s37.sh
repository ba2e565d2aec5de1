# fold x tasks with an extension record (raw plane pointers, first child's plane pointers, Ik base, number of children) fetched beside the task descriptor: parity (fold / root / DT / batch / group
# tests, float + double, compact plans), per-phase trace of a fold launch, A/B against commit ebad0f4's build (prev)
mkdir -p gpurun_out/r06_s37
timeout 1300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_f64.py -x -q -m gpu -k "dt2d or dp_min or detect or person or fuzz or dp_pointers or trees or chains or mixtures or configs0 or 1080p or f64 or root or nms or group or batch or argmin or compact or tune" > gpurun_out/r06_s37/pytest_dt.log 2>&1
tail -2 gpurun_out/r06_s37/pytest_dt.log | cut -c1-200
python tests/tools_dt_trace.py 640 480 2 8 2>&1 | grep "batch of 8"
python tests/tools_dt_trace.py 640 480 2 2>&1 | grep -E "^(0|256|512|768) mean"
bash profiles/r06/sessions/ab.sh r06_s37 4 libpbd_hip_prev.so libpbd_hip.so
