# per-launch durations of k_dt_pass / k_reduce inside a frame (sequential frames): bash profiles/dt_launches.sh
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/dtl
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python $REPO/bench.py --graph 0 --no-prewarm --steps 6 --warmup 2 --inflight 1 --no-cpu-baseline > $OUT/log 2>&1
python - <<'PY'
import csv, glob, os
out=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/dtl"
f=glob.glob(out+"/**/*kernel_trace.csv", recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
frames=[]; cur=[]
for r in rows:
    k=r["Kernel_Name"]
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    if "k_hog" in k:
        if cur: frames.append(cur)
        cur=[]
    if "k_dt_pass" in k or "k_reduce" in k or "k_root" in k: cur.append((k.split("<")[0].replace("void ",""), d, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
if cur: frames.append(cur)
frames=[fr for fr in frames if len(fr)==len(frames[-1])][-4:]
n=len(frames[0])
tot=0
for i in range(n):
    ds=[fr[i][1] for fr in frames]
    gap=[(fr[i][2]-fr[i-1][3])/1e3 for fr in frames] if i else [0]
    tot+=sum(ds)/len(ds)
    print(i, frames[0][i][0], "%.1f us" % (sum(ds)/len(ds)), "gap before %.1f" % (sum(gap)/len(gap)))
print("sum of kernels %.1f us; first start to last end %.1f us" % (tot, sum((fr[-1][3]-fr[0][2])/1e3 for fr in frames)/len(frames)))
PY
find $OUT -name "*.csv" -delete
