set -u
O=gpurun_out/r06_s1; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_driverflags.json 2> $O/bench.err
for l in 0 1 2; do
  python tests/tools_dt_trace.py 640 480 $l > $O/trace_single_l$l.txt 2>&1
  python tests/tools_dt_trace.py 640 480 $l 16 > $O/trace_b16_l$l.txt 2>&1
done
PBD_LIBRARY=$PWD/partsbaseddetector_amd/libpbd_hip_tune.so PBD_DEBUG_PLAN=1 python -c "
from partsbaseddetector_amd import capi
from partsbaseddetector_amd.model import make_image, make_person_model
m = make_person_model(K=6); m.thresh=1e9
h = capi.Handle(m, graph=0)
h.detect(make_image(0,640,480))
" > $O/plan.txt 2>&1
tail -c 1500 $O/bench_driverflags.json
