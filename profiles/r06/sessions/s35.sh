# dt_stitch1: the intersection's operands (x of both entries, 1/dx) read with the iteration's other LDS reads (one round trip per iteration instead of two): A/B, 4 interleaved runs; DT tests on the variant
mkdir -p gpurun_out/r06_s35
PBD_LIBRARY=$PWD/partsbaseddetector_amd/libpbd_hip_pin.so timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_f64.py -x -q -m gpu -k "dt2d or dp_min or detect_exact or person_full_size or fuzz or dp_pointers or trees or chains or mixtures or configs0 or 1080p or f64" > gpurun_out/r06_s35/pytest_dt.log 2>&1
tail -2 gpurun_out/r06_s35/pytest_dt.log | cut -c1-200
bash profiles/r06/sessions/ab.sh r06_s35 4 libpbd_hip.so libpbd_hip_pin.so
