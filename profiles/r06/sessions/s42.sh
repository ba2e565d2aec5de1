# image pyramid kernels: the BGR resize's four pixels' loads in flight together (no branch on the row's last pixel), pyrDown's window words fetched together (reflected words in a second pass):
# parity (pyramid / HOG / image / detect tests, float + double), A/B against the previous build
mkdir -p gpurun_out/r06_s42
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_f64.py -x -q -m gpu -k "hog or pyramid or pyrdown or resize or wide or image or detect_exact or person_full_size or fuzz or configs0 or feature or 1080p" > gpurun_out/r06_s42/pytest_pyr.log 2>&1
tail -2 gpurun_out/r06_s42/pytest_pyr.log | cut -c1-200
bash profiles/r06/sessions/ab.sh r06_s42 4 libpbd_hip_prev.so libpbd_hip.so
python - <<'PY'
import json, glob
for L in ("libpbd_hip_prev", "libpbd_hip"):
    for f in sorted(glob.glob(f"gpurun_out/r06_s42/{L}_[0-9].json")):
        d = json.load(open(f))
        print(L, 'pyramid batched', d["stage_ms_per_frame_batched"]["image_pyramid"], 'alone', d["stage_ms_sequential"]["image_pyramid"], 'value', d['value'], 'lat', d['sequential']['latency_ms']['median'])
PY
