#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect.sh r01c
# writes gpurun_out/<tag>/{stats_seq,stats,pmc_fetch,pmc_write,pmc_clk}/ + bench JSON lines; afterwards
#   python profiles/summarize.py gpurun_out/<tag> <tag>     (here, on the merged gpurun_out)
# turns them into profiles/<tag>_*.  Counters run in their own passes (--pmc with --kernel-trace only).
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
B="python $REPO/bench.py --graph 0 --no-prewarm --batch 1"   # eager launches, single frames under the profiler (the timed loop of the default run replays a hipGraph of a batch)
# throughput line (default flags) and the sequential line
python $REPO/bench.py --steps 300 > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_n1_driverflags.json" 2>> "$OUT/bench_n1.err"
$B --steps 100 --inflight 1 --no-cpu-baseline > "$OUT/bench_n1_inflight1.json" 2>> "$OUT/bench_n1.err"
python $REPO/bench.py --steps 300 --inflight 4 --batch 1 --no-cpu-baseline > "$OUT/bench_n1_b1.json" 2>> "$OUT/bench_n1.err"
for sb in "4 3" "3 4" "4 4" "2 8"; do set -- $sb; python $REPO/bench.py --steps 60 --inflight $1 --batch $2 --no-cpu-baseline 2>> "$OUT/bench_n1.err" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('inflight $1 batch $2:', d['value'], 'frames/s')" >> "$OUT/sweep_sb.txt"; done
for sb in "4 3" "3 4" "4 4"; do set -- $sb; python $REPO/bench.py --steps 20 --warmup 5 --inflight $1 --batch $2 --no-cpu-baseline 2>> "$OUT/bench_n1.err" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('steps 20 warmup 5: inflight $1 batch $2:', d['value'], 'frames/s')" >> "$OUT/sweep_sb.txt"; done
python $REPO/tests/tools_batch_stages.py 1 2 4 8 > "$OUT/batch_stages.txt" 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_batch" -o run -- python $REPO/bench.py --graph 0 --no-prewarm --steps 12 --warmup 3 --inflight 1 --batch 4 --no-cpu-baseline > "$OUT/stats_batch.log" 2>&1
$B --steps 50 --dtype f64 > "$OUT/bench_n1_f64.json" 2>> "$OUT/bench_n1.err"
# per-kernel durations: sequential (undisturbed) and default (4 frames in flight)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_seq" -o run -- $B --steps 20 --warmup 3 --inflight 1 --no-cpu-baseline > "$OUT/stats_seq.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- $B --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_f64" -o run -- $B --steps 10 --warmup 2 --inflight 1 --dtype f64 --no-cpu-baseline > "$OUT/stats_f64.log" 2>&1
# HBM traffic and clock: one counter per pass
for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_$c" -o run -- $B --steps 4 --warmup 2 --inflight 1 --no-cpu-baseline > "$OUT/pmc_$c.log" 2>&1
done
# the same two counters on the benched unit of work: one launch chain per batch of BATCHN frames (bench.py's default), batches one at a time
BATCHN=${BATCHN:-8}
echo $BATCHN > "$OUT/batch_n.txt"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmcb_$c" -o run -- python $REPO/bench.py --graph 0 --no-prewarm --batch $BATCHN --steps 4 --warmup 2 --inflight 1 --no-cpu-baseline > "$OUT/pmcb_$c.log" 2>&1
done
# SQ activity of the distance-transform / reduce / filter-bank kernels: one counter per pass
for c in SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/sq_$c" -o run -- $B --steps 4 --warmup 2 --inflight 1 --no-cpu-baseline > "$OUT/sq_$c.log" 2>&1
done
# configs[4]: direct VALU correlation vs MFMA implicit GEMM for N = 26 .. 312 filters (stage times + per-kernel view)
if [ -z "${SKIP_CONV_MODES:-}" ]; then   # unchanged filter bank: SKIP_CONV_MODES=1 keeps the previous collection's table
python $REPO/profiles/conv_modes.py > "$OUT/conv_modes.json" 2> "$OUT/conv_modes.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_conv_modes" -o run -- python $REPO/profiles/conv_modes.py > /dev/null 2>&1
fi
# gpurun merges at most 64 MiB back: the per-dispatch traces are not needed by summarize.py (stats + counter CSVs are)
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*agent_info.csv" -delete
du -sh "$OUT"
