# tiles carrying a copy of their level descriptor (k_hog, k_conv_split32: one scalar round trip at the top of the workgroup instead of tile -> level): parity, A/B against the previous build
mkdir -p gpurun_out/r06_s49
timeout 1300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_f64.py -x -q -m gpu -k "hog or pdf or split or conv or image or detect_exact or person_full_size or fuzz or configs0 or feature or batch or mfma" > gpurun_out/r06_s49/pytest.log 2>&1
tail -2 gpurun_out/r06_s49/pytest.log | cut -c1-200
bash profiles/r06/sessions/ab.sh r06_s49 4 libpbd_hip_prev.so libpbd_hip.so
python - <<'PY'
import json, glob
for L in ("libpbd_hip_prev", "libpbd_hip"):
    for f in sorted(glob.glob(f"gpurun_out/r06_s49/{L}_[0-9].json")):
        d = json.load(open(f))
        print(L, 'hog', d["stage_ms_per_frame_batched"]["hog"], d["stage_ms_sequential"]["hog"], 'pdf', d["stage_ms_per_frame_batched"]["pdf"], d["stage_ms_sequential"]["pdf"], 'value', d['value'], 'lat', d['sequential']['latency_ms']['median'])
PY
