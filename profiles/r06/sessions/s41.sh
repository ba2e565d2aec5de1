# (a) kernel trace of the TIMED leg on the final tree reduced to its concurrency (profiles/overlap_trace.py; VERDICT r05 #3 asks for the round's overlap trace)
# (b) 1920x1080 / 1280x720: the two float DT geometries once more on the final kernels (tuning build); (c) handles in flight x frames per batch on the final kernels
set -u
O=$PWD/gpurun_out/r06_s41; mkdir -p $O
REPO=$PWD
( cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --output-format csv -d $O/trace_timed -o run -- python $REPO/bench.py --steps 30 --legs timed > $O/bench_timed.json 2> $O/bench_timed.err
  python $REPO/profiles/overlap_trace.py $O/trace_timed > $O/overlap_timed.json 2> $O/overlap_timed.err
  find $O/trace_timed -name "*kernel_trace.csv" -delete; find $O/trace_timed -name "*agent_info.csv" -delete )
cat $O/overlap_timed.json | head -30
export PBD_LIBRARY=$PWD/partsbaseddetector_amd/libpbd_hip_tune.so
for sz in "1920 1080" "1280 720"; do
  set -- $sz
  for geo in dflt big; do
    if [ $geo = big ]; then export PBD_DT_NT=256 PBD_DT_NT_X=256 PBD_DT_BUDGET_KB=40 PBD_DT_BUDGET_X_KB=40; else unset PBD_DT_NT PBD_DT_NT_X PBD_DT_BUDGET_KB PBD_DT_BUDGET_X_KB; fi
    python bench.py --width $1 --height $2 --steps 8 --warmup 2 --legs timed,batchseq,seq > $O/${1}_$geo.json 2>> $O/err.log
  done
done
unset PBD_DT_NT PBD_DT_NT_X PBD_DT_BUDGET_KB PBD_DT_BUDGET_X_KB PBD_LIBRARY
for sb in "3 16" "4 16" "3 24" "4 12" "2 24"; do
  set -- $sb
  for r in 1 2; do python bench.py --steps 40 --warmup 5 --legs timed --inflight $1 --batch $2 > $O/sb_${1}x${2}_$r.json 2>> $O/err.log; done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_s41/1*_*.json')):
    d = json.load(open(f))
    print(f.split('/')[-1], 'value', d['value'], 'dp batched', d['stage_ms_per_frame_batched']['dp_min'], 'dp alone', d['stage_ms_sequential']['dp_min'], 'frac', d['roofline']['frac'])
for f in sorted(glob.glob('gpurun_out/r06_s41/sb_*.json')):
    d = json.load(open(f))
    print(f.split('/')[-1], 'value', d['value'])
PY
