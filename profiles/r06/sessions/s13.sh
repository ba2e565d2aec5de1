set -u
O=gpurun_out/r06_s13; mkdir -p $O
export PBD_LIBRARY=$PWD/partsbaseddetector_amd/libpbd_hip_tune.so
run() { # name, env...
  n=$1; shift
  env "$@" python bench.py --steps 40 --warmup 5 --legs timed,batchseq,seq > $O/$n.json 2>> $O/err.log
  python - $O/$n.json $n <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"{sys.argv[2]:28s} value {d['value']:.1f} dpb {d['stage_ms_per_frame_batched']['dp_min']:.4f} dps {d['stage_ms_sequential']['dp_min']:.4f} lat {d['sequential']['latency_ms']['median']:.3f}")
PY
}
run dflt_a X=1
run kb36 PBD_DT_BUDGET_KB=36 PBD_DT_BUDGET_X_KB=36
run kb44 PBD_DT_BUDGET_KB=44 PBD_DT_BUDGET_X_KB=44
run kb52 PBD_DT_BUDGET_KB=52 PBD_DT_BUDGET_X_KB=52
run x36 PBD_DT_BUDGET_X_KB=36
run x48 PBD_DT_BUDGET_X_KB=48
run nt192 PBD_DT_NT=192 PBD_DT_NT_X=192 PBD_DT_BUDGET_KB=40 PBD_DT_BUDGET_X_KB=40
run nt192_30 PBD_DT_NT=192 PBD_DT_NT_X=192 PBD_DT_BUDGET_KB=30 PBD_DT_BUDGET_X_KB=30
run seg12 PBD_DT_SEG=12
run seg20 PBD_DT_SEG=20
run xcd8 PBD_DT_XCD_CHUNK=8
run xcd32 PBD_DT_XCD_CHUNK=32
run dflt_b X=1
