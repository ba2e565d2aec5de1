#!/bin/bash
# session 39: full GPU suite + smoke on the final code
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO; OUT=$REPO/gpurun_out/r03ah; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 > $OUT/pytest_all.log 2>&1; echo "pytest rc=$?" > $OUT/summary.txt
grep -E "passed|failed" $OUT/pytest_all.log | tail -2 >> $OUT/summary.txt
python -c "import __graft_entry__ as g; g.smoke()" >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt | tail -6
