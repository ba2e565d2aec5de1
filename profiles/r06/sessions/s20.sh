timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hog or pdf_exact or pdf_split_products_tolerance or dt2d_bit_exact or dp_min or detect_exact or person_full_size or batch_equals or nms_map or resize or pyrdown" 2>&1 | tail -2
bash profiles/r06/sessions/ab.sh r06_s20 3 libpbd_hip_v5.so libpbd_hip.so
python - <<'PY'
import json, glob
for L in ("libpbd_hip_v5", "libpbd_hip"):
    for f in sorted(glob.glob(f"gpurun_out/r06_s20/{L}_[0-9].json")):
        d = json.load(open(f))
        print(L, d["stage_ms_sequential"])
PY
