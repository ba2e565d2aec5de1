set -u
O=gpurun_out/r06_s6; mkdir -p $O
python - <<'PY'
import numpy as np
from partsbaseddetector_amd import capi
from partsbaseddetector_amd.model import make_image, make_person_model
m = make_person_model(K=6); m.thresh = 0.0
im = make_image(3, 640, 480)
ref = None
for G in (1, 2, 4, 8):
    for graph in (0, 1):
        h = capi.Handle(m, conv_mode=capi.PBD_CONV_EXACT, graph=graph)
        h.set_dp_level_groups(G)
        outs = [h.detect(im, capacity=65536) for _ in range(3)]
        assert h.dp_level_groups == G, (h.dp_level_groups, G)
        for o in outs:
            if ref is None: ref = o
            assert len(o[0]) == len(ref[0]) and np.array_equal(o[0]["score"], ref[0]["score"]) and np.array_equal(o[1], ref[1]) and np.array_equal(o[2], ref[2]), (G, graph)
        h.close()
print("level groups 1/2/4/8 x eager/graph: identical candidates", len(ref[0]))
PY
for G in 1 2 3 4 6 8; do
  python bench.py --steps 40 --warmup 5 --legs seq,single --dp-groups $G > $O/g$G.json 2>> $O/err.log
done
for G in 1 4; do
  python bench.py --steps 40 --warmup 5 --legs timed,seq --batch 1 --inflight 4 --dp-groups $G > $O/b1_g$G.json 2>> $O/err.log
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_s6/*.json')):
    d = json.load(open(f))
    print(f.split('/')[-1], 'groups', d['config'].get('dp_level_groups_single_frames'), 'lat', d['sequential']['latency_ms'], 'dps', d['stage_ms_sequential']['dp_min'], 'single', d.get('value_single_frame_calls'), 'value', d.get('value'))
PY
