# single frames: HOG of the first octave on a forked stream beside the pyrDown chain (fork / join inside the hipGraph): parity on the single-frame detect paths, A/B against the previous build
mkdir -p gpurun_out/r06_s46
timeout 1400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_f64.py -x -q -m gpu -k "detect or person or async or graph or fuzz or configs0 or group or levels or stream or stage or nms or image or compact or tune" > gpurun_out/r06_s46/pytest.log 2>&1
tail -2 gpurun_out/r06_s46/pytest.log | cut -c1-200
bash profiles/r06/sessions/ab.sh r06_s46 4 libpbd_hip_prev.so libpbd_hip.so
