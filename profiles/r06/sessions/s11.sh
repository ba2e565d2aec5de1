timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_f64.py -x -q -m gpu -k "hog or pyramid or detect_exact or person_full_size or wide_images or random_sizes or face_like" 2>&1 | tail -3
bash profiles/r06/sessions/ab.sh r06_s11 3 libpbd_hip_v3.so libpbd_hip.so
python - <<'PY'
import json, glob
for L in ("libpbd_hip_v3", "libpbd_hip"):
    for f in sorted(glob.glob(f"gpurun_out/r06_s11/{L}_[0-9].json")):
        d = json.load(open(f))
        print(L, d["stage_ms_per_frame_batched"], d["stage_ms_sequential"]["hog"])
PY
