# (a) read-out with the piece below the current one (and its link) kept in registers: a step down needs no LDS wait (rounds 1-3 had it; round 4 removed it for four vector instructions per step)
# (b) k_hog's gradient phase with 2 / 4 pixels per thread in flight, unpredicated, the channel count at compile time
mkdir -p gpurun_out/r06_s39
PBD_LIBRARY=$PWD/partsbaseddetector_amd/libpbd_hip_ra.so timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dt2d or dp_min or detect_exact or person_full_size or fuzz or dp_pointers or trees or chains or mixtures or configs0" > gpurun_out/r06_s39/pytest_dt.log 2>&1
tail -2 gpurun_out/r06_s39/pytest_dt.log | cut -c1-200
PBD_LIBRARY=$PWD/partsbaseddetector_amd/libpbd_hip_g4.so timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hog or pyramid or image or detect_exact or person_full_size or fuzz or configs0 or feature" > gpurun_out/r06_s39/pytest_hog.log 2>&1
tail -2 gpurun_out/r06_s39/pytest_hog.log | cut -c1-200
bash profiles/r06/sessions/ab.sh r06_s39 4 libpbd_hip.so libpbd_hip_ra.so libpbd_hip_g2.so libpbd_hip_g4.so
python - <<'PY'
import json, glob
for L in ("libpbd_hip", "libpbd_hip_g2", "libpbd_hip_g4"):
    for f in sorted(glob.glob(f"gpurun_out/r06_s39/{L}_[0-9].json")):
        d = json.load(open(f))
        print(L, 'hog batched', d["stage_ms_per_frame_batched"]["hog"], 'alone', d["stage_ms_sequential"]["hog"], 'value', d['value'])
PY
