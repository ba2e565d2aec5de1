set -u
O=gpurun_out/r06_s10; mkdir -p $O
export PBD_LIBRARY=$PWD/partsbaseddetector_amd/libpbd_hip_tune.so
for sz in "1920 1080" "1280 720"; do
  set -- $sz
  for geo in dflt big; do
    if [ $geo = big ]; then export PBD_DT_NT=256 PBD_DT_NT_X=256 PBD_DT_BUDGET_KB=40 PBD_DT_BUDGET_X_KB=40; else unset PBD_DT_NT PBD_DT_NT_X PBD_DT_BUDGET_KB PBD_DT_BUDGET_X_KB; fi
    python bench.py --width $1 --height $2 --steps 8 --warmup 2 --legs timed,batchseq,seq > $O/${1}_$geo.json 2>> $O/err.log
  done
done
unset PBD_DT_NT PBD_DT_NT_X PBD_DT_BUDGET_KB PBD_DT_BUDGET_X_KB
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_s10/*.json')):
    d = json.load(open(f))
    print(f.split('/')[-1], 'value', d['value'], 'dp batched', d['stage_ms_per_frame_batched']['dp_min'], 'dp alone', d['stage_ms_sequential']['dp_min'], 'frac', d['roofline']['frac'], 'stages', d['stage_ms_per_frame_batched'])
PY
