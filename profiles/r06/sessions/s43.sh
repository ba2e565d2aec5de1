# the split bank's tuning variants once more, now that the DT runs at issue priority 1 in front of it (tuning build, timed leg, 2 runs each): 0 = default, 2 = two-wavefront workgroups, 7 / 8 = n-tile groups of four / three
set -u
O=gpurun_out/r06_s43; mkdir -p $O
export PBD_LIBRARY=$PWD/partsbaseddetector_amd/libpbd_hip_tune.so
for r in 1 2; do for v in 0 2 7 8 4; do
  PBD_SPLIT_VARIANT=$v python bench.py --steps 40 --warmup 5 --legs timed,batchseq > $O/v${v}_$r.json 2>> $O/err.log
done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_s43/v*.json')):
    d = json.load(open(f)); print(f.split('/')[-1], 'value', d['value'], 'pdf', d['stage_ms_per_frame_batched']['pdf'], 'dp', d['stage_ms_per_frame_batched']['dp_min'])
PY
