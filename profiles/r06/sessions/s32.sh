# k_hog: branch-free staging of the source pixels (7 loads in flight per thread instead of one): parity on the HOG / image tests (float + double), per-phase stamps, A/B against the previous kernel
mkdir -p gpurun_out/r06_s32
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_f64.py -x -q -m gpu -k "hog or pyramid or wide or image or detect_exact or person_full_size or fuzz or configs0 or feature" > gpurun_out/r06_s32/pytest_hog.log 2>&1
tail -2 gpurun_out/r06_s32/pytest_hog.log | cut -c1-200
python tests/tools_hog_probe.py 2>&1 | grep phases | tail -1
bash profiles/r06/sessions/ab.sh r06_s32 3 libpbd_hip_hogold.so libpbd_hip.so
python - <<'PY'
import json, glob
for L in ("libpbd_hip_hogold", "libpbd_hip"):
    for f in sorted(glob.glob(f"gpurun_out/r06_s32/{L}_[0-9].json")):
        d = json.load(open(f))
        print(L, 'hog batched', d["stage_ms_per_frame_batched"]["hog"], 'alone', d["stage_ms_sequential"]["hog"], 'pyramid', d["stage_ms_per_frame_batched"]["pyramid"], d["stage_ms_sequential"]["pyramid"])
PY
