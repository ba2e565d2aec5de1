#!/bin/bash
# session 32: full GPU suite on the product library + the round's final rocprofv3 collection (r03c: 3 handles x batches of 8)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
mkdir -p gpurun_out/r03c
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/r03c/pytest_all.log 2>&1; echo "pytest rc=$?" > gpurun_out/r03c/summary.txt
tail -3 gpurun_out/r03c/pytest_all.log >> gpurun_out/r03c/summary.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/r03c/summary.txt 2>&1
SKIP_CONV_MODES=1 bash profiles/collect.sh r03c > gpurun_out/collect_r03c.log 2>&1
cat gpurun_out/r03c/summary.txt gpurun_out/r03c/sweep_sb.txt
tail -2 gpurun_out/collect_r03c.log
