# k_hog without the block-index tables in the even-cell instantiations: 53 424 B of LDS instead of 54 032 (three workgroups per CU instead of two, if LDS is allocated in 1 280-byte granules):
# parity (HOG / image / detect tests, float + double, + the new dt2d test), A/B against the previous build
mkdir -p gpurun_out/r06_s44
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_f64.py -x -q -m gpu -k "hog or pyramid or wide or image or detect_exact or person_full_size or fuzz or configs0 or feature or fused_and_unfused or sbin" > gpurun_out/r06_s44/pytest_hog.log 2>&1
tail -2 gpurun_out/r06_s44/pytest_hog.log | cut -c1-200
bash profiles/r06/sessions/ab.sh r06_s44 4 libpbd_hip_prev.so libpbd_hip.so
python - <<'PY'
import json, glob
for L in ("libpbd_hip_prev", "libpbd_hip"):
    for f in sorted(glob.glob(f"gpurun_out/r06_s44/{L}_[0-9].json")):
        d = json.load(open(f))
        print(L, 'hog batched', d["stage_ms_per_frame_batched"]["hog"], 'alone', d["stage_ms_sequential"]["hog"], 'value', d['value'], 'lat', d['sequential']['latency_ms']['median'], 'single', d['value_single_frame_calls'])
PY
