timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "detect_exact or person_full_size or batch or group or appends_and_capacity or async or graph_replay or argmin or nms_person or fuzz_detect" 2>&1 | tail -2
bash profiles/r06/sessions/ab.sh r06_s26 3 libpbd_hip_nz.so libpbd_hip.so
python - <<'PY'
import json, glob
for L in ("libpbd_hip_nz", "libpbd_hip"):
    for f in sorted(glob.glob(f"gpurun_out/r06_s26/{L}_[0-9].json")):
        d = json.load(open(f))
        print(L, 'argmin seq', d["stage_ms_sequential"]["argmin"], 'batched', d["stage_ms_per_frame_batched"]["argmin"], 'h2d', d.get("value_incl_h2d"))
PY
