# k_hog staging, second form (7 loads in flight at 31 registers): parity on the HOG / image tests, A/B against the previous kernel (5 interleaved runs)
mkdir -p gpurun_out/r06_s33
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_f64.py -x -q -m gpu -k "hog or pyramid or wide or image or detect_exact or person_full_size or fuzz or configs0 or feature" > gpurun_out/r06_s33/pytest_hog.log 2>&1
tail -2 gpurun_out/r06_s33/pytest_hog.log | cut -c1-200
python tests/tools_hog_probe.py 2>&1 | grep phases | tail -1
bash profiles/r06/sessions/ab.sh r06_s33 5 libpbd_hip_hogold.so libpbd_hip.so
python - <<'PY'
import json, glob
for L in ("libpbd_hip_hogold", "libpbd_hip"):
    for f in sorted(glob.glob(f"gpurun_out/r06_s33/{L}_[0-9].json")):
        d = json.load(open(f))
        print(L, 'hog batched', d["stage_ms_per_frame_batched"]["hog"], 'alone', d["stage_ms_sequential"]["hog"], 'value', d['value'])
PY
