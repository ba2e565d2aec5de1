#!/bin/bash
# r05 session 12: kernel trace of the TIMED leg (three batches in flight, graph replay) reduced to the concurrency it shows (profiles/overlap_trace.py),
# for the split bank and for the fp32 MFMA bank
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r05s12; mkdir -p $O
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for conv in auto mfma; do
  rocprofv3 --kernel-trace --output-format csv -d $O/trace_$conv -o run -- python $REPO/bench.py --steps 30 --legs timed --conv $conv > $O/bench_$conv.json 2> $O/bench_$conv.err
  python $REPO/profiles/overlap_trace.py $O/trace_$conv > $O/overlap_$conv.json 2> $O/overlap_$conv.err
  find $O/trace_$conv -name "*kernel_trace.csv" -delete; find $O/trace_$conv -name "*agent_info.csv" -delete
  echo "== $conv"; cat $O/overlap_$conv.json; python -c "import json,sys; d=json.loads(open('$O/bench_$conv.json').read().strip().splitlines()[-1]); print('value under the profiler', d['value'])"
done
