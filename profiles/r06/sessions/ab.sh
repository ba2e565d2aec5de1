# A/B of library builds on one box, alternating runs: bash profiles/r06/sessions/ab.sh <tag> <reps> <libA> <libB> [...]
# (names relative to partsbaseddetector_amd/, e.g. libpbd_hip_base.so libpbd_hip.so); prints value / dp_min batched / dp_min alone / latency
set -u
TAG=$1; REPS=$2; shift 2
O=gpurun_out/$TAG; mkdir -p $O
for r in $(seq 1 $REPS); do
  for L in "$@"; do
    PBD_LIBRARY=$PWD/partsbaseddetector_amd/$L python bench.py --steps 60 --warmup 5 --legs timed,batchseq,seq,single > $O/${L%.so}_$r.json 2>> $O/err.log
  done
done
python - "$O" "$@" <<'PY'
import json, sys, glob
O = sys.argv[1]
for L in sys.argv[2:]:
    rows = []
    for f in sorted(glob.glob(f"{O}/{L[:-3]}_[0-9]*.json")):
        try:
            d = json.load(open(f))
        except Exception as e:
            print(L, f, "unreadable", e); continue
        rows.append((d["value"], d["stage_ms_per_frame_batched"]["dp_min"], d["stage_ms_sequential"]["dp_min"], d["sequential"]["latency_ms"]["median"],
                     d["roofline"]["frac"], d.get("value_single_frame_calls"), d["stage_ms_per_frame_batched"]["pdf"]))
    print(f"{L:28s}", " | ".join(f"{v:.1f} dpb {b:.4f} dps {s:.4f} lat {l:.3f} frac {fr:.4f} single {sf:.0f} pdf {p:.4f}" for v, b, s, l, fr, sf, p in rows))
PY
