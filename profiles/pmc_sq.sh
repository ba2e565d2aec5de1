# SQ counter passes (one counter per run) for the DT / filter-bank / reduce kernels: bash profiles/pmc_sq.sh (through gpurun)
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_dt2
mkdir -p $OUT
for c in SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o run -- python $REPO/bench.py --steps 4 --warmup 2 --inflight 1 --no-cpu-baseline > $OUT/$c.log 2>&1
done
python - <<'PY'
import csv, glob, os
out=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/pmc_dt2"
for c in sorted(os.listdir(out)):
    f=glob.glob(f"{out}/{c}/*counter_collection.csv")
    if not f: continue
    acc={}
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"]!=c: continue
        k=r["Kernel_Name"].split("(")[0]
        a=acc.setdefault(k,[0,0.0,0]); a[0]+=1; a[1]+=float(r["Counter_Value"]); a[2]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
    for k,a in acc.items():
        if "k_dt_pass" in k or "k_conv" in k or "k_reduce" in k: print(c, k[:40], "calls",a[0],"per call %.3g"%(a[1]/a[0]), "per us %.4g"%(a[1]/(a[2]/1e3)))
PY
