#!/bin/bash
# Round-5 evidence collection on the GPU box (through gpurun, from the repo root), in PARTS so that a session can take what it
# has GPU minutes for:
#   bash profiles/collect_r05.sh <tag> [part ...]        parts: bench trace8 traceseq pmc8 sq sq2 sq3 trace1080 f64   (default: all but trace1080 / f64)
# writes gpurun_out/<tag>/...; the per-dispatch kernel traces are reduced ON THE BOX (profiles/reduce_trace.py) and deleted
# (gpurun merges at most 64 MiB back).  Afterwards, here:  python profiles/summarize_r05.py gpurun_out/<tag> <tag>
#
#   trace8    rocprofv3 --kernel-trace --stats of a run that launches ONLY launch chains of the benched unit — batches of 8
#             frames, one at a time, eager (bench.py --legs batchseq): the same leg the bench line's `roofline.launch_ms`
#             comes from (HIP events), so that the headline roofline can be recomputed from profiles/<tag>_kernel_stats_batch8.csv
#   traceseq  the same for single frames one at a time (`roofline_single_frame`)
#   pmc8      FETCH_SIZE / WRITE_SIZE of the batch-of-8 chains (separate passes; profiles/traffic_dp.json)
#   sq        SQ counters of the same chains: ONE pass for eight SQ counters (the SQ block has 8 slots, MI355X_MICROARCH.md)
#   sq2, sq3  further SQ passes: active lanes per vector instruction (THREAD_CYCLES_VALU / ACTIVE_INST_VALU); matrix-pipe busy + MFMA ops of the split bank
# Counter passes use --pmc with --kernel-trace only (no other trace domain).
set -u
TAG=${1:-r05}
shift || true
PARTS=${*:-bench trace8 traceseq pmc8 sq}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py"
B8="$BENCH --legs batchseq --graph 0 --inflight 1 --no-prewarm --warmup 2"          # only launch chains of the benched batch size (bench.py's default) (+ the one single-frame chain of the threshold pick)
B1="$BENCH --legs seq --graph 0 --inflight 1 --no-prewarm --warmup 2 --batch 1"      # only single-frame chains
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
reduce() {  # <dir> <out prefix>: per-(kernel, grid) stats + dp_min chains, then drop the big per-dispatch files
  python $REPO/profiles/reduce_trace.py "$1" "$2" > "$2_reduce.log" 2>&1
  find "$1" -name "*kernel_trace.csv" -delete; find "$1" -name "*agent_info.csv" -delete
}
if has bench; then
  $BENCH --steps 300 > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
  $BENCH --steps 20 --warmup 5 > "$OUT/bench_n1_driverflags.json" 2>> "$OUT/bench_n1.err"
  $BENCH --steps 300 --inflight 4 --batch 1 --legs timed > "$OUT/bench_n1_b1.json" 2>> "$OUT/bench_n1.err"
  python $REPO/tests/tools_batch_stages.py 1 2 4 8 16 > "$OUT/batch_stages.txt" 2>/dev/null
fi
if has trace8; then
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace8" -o run -- $B8 > "$OUT/trace8.json" 2> "$OUT/trace8.err"
  reduce "$OUT/trace8" "$OUT/batch8"
fi
if has traceseq; then
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/traceseq" -o run -- $B1 > "$OUT/traceseq.json" 2> "$OUT/traceseq.err"
  reduce "$OUT/traceseq" "$OUT/seq"
fi
if has pmc8; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc8_$c" -o run -- $B8 > "$OUT/pmc8_$c.log" 2>&1
    find "$OUT/pmc8_$c" -name "*kernel_trace.csv" -delete; find "$OUT/pmc8_$c" -name "*agent_info.csv" -delete
  done
fi
if has sq; then
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT \
    --kernel-trace --output-format csv -d "$OUT/sq8" -o run -- $B8 > "$OUT/sq8.log" 2>&1
  find "$OUT/sq8" -name "*kernel_trace.csv" -delete; find "$OUT/sq8" -name "*agent_info.csv" -delete
fi
if has sq2; then   # lane utilisation: active lanes per vector instruction = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU (of 64)
  rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
    --kernel-trace --output-format csv -d "$OUT/sq8b" -o run -- $B8 > "$OUT/sq8b.log" 2>&1
  find "$OUT/sq8b" -name "*kernel_trace.csv" -delete; find "$OUT/sq8b" -name "*agent_info.csv" -delete
fi
if has sq3; then   # matrix pipe of the split-product bank
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_BUSY_CYCLES \
    --kernel-trace --output-format csv -d "$OUT/sq8c" -o run -- $B8 > "$OUT/sq8c.log" 2>&1
  find "$OUT/sq8c" -name "*kernel_trace.csv" -delete; find "$OUT/sq8c" -name "*agent_info.csv" -delete
fi
if has pmcpdf; then   # HBM traffic incl. the filter bank's (all kernels of the batch chains are in the same passes as pmc8: per-kernel table)
  :
fi
if has trace1080; then
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace1080" -o run -- $BENCH --legs seq --graph 0 --inflight 1 --no-prewarm --warmup 2 --batch 1 --width 1920 --height 1080 > "$OUT/trace1080.json" 2> "$OUT/trace1080.err"
  reduce "$OUT/trace1080" "$OUT/seq1080"
fi
if has f64; then
  $BENCH --steps 50 --dtype f64 --legs timed,seq,batchseq > "$OUT/bench_n1_f64.json" 2>> "$OUT/bench_n1.err"
fi
du -sh "$OUT"
