#!/bin/bash
# r04 session 2: full GPU suite (session 1 stopped at the stale tuning library)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s2
timeout 1200 python -m pytest tests -m gpu -x -q -s > gpurun_out/r04s2/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04s2/pytest.log
grep -E "MFMA bank vs oracle|person 26x6|passed|failed|rc=" gpurun_out/r04s2/pytest.log | tail -12
