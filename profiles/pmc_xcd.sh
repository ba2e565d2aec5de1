# HBM traffic and time of the DT passes against the XCD chunking of the task table (probe build):
# bash profiles/pmc_xcd.sh (through gpurun)
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_xcd
mkdir -p $OUT
export PBD_LIBRARY=$REPO/partsbaseddetector_amd/libpbd_hip_probes.so
for k in ${CHUNKS:-0 1 2 4 8 16}; do
  export PBD_DT_XCD_CHUNK=$k
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$k/$c -o run -- python $REPO/bench.py --steps 4 --warmup 2 --inflight 1 --graph 0 --no-prewarm --no-cpu-baseline > $OUT/$k.$c.log 2>&1
  done
  python $REPO/bench.py --steps 40 --inflight 1 --no-cpu-baseline > $OUT/$k.seq.json 2>/dev/null
  python $REPO/bench.py --steps 200 --no-cpu-baseline > $OUT/$k.thr.json 2>/dev/null
  find $OUT/$k -name '*kernel_trace.csv' -delete; find $OUT/$k -name '*agent_info.csv' -delete
done
python - <<'PY'
import csv, glob, os, json
out=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/pmc_xcd"
for k in sorted([d for d in os.listdir(out) if os.path.isdir(out+"/"+d)], key=int):
    row={}
    for c in ("FETCH_SIZE","WRITE_SIZE"):
        f=glob.glob(f"{out}/{k}/{c}/*counter_collection.csv")
        if not f: continue
        acc={}; nfr=0
        for r in csv.DictReader(open(f[0])):
            if r["Counter_Name"]!=c: continue
            kn=r["Kernel_Name"].split("(")[0]
            acc[kn]=acc.get(kn,0.0)+float(r["Counter_Value"])
            nfr+="k_root" in kn            # one k_root launch per frame
        for kn,v in acc.items():
            if "k_dt_pass" in kn or "k_reduce" in kn: row[(c,kn[:20])]=v/max(nfr,1)   # KB per frame
    s=json.load(open(f"{out}/{k}.seq.json")); t=json.load(open(f"{out}/{k}.thr.json"))
    print("chunk",k, {f"{a}:{b}":"%.1f MB"%(v*1e3/1e6) for (a,b),v in row.items()}, "dp_min ms", s["stage_ms_sequential"]["dp_min"], "fps", t["value"])
PY
