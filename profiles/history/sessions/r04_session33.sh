#!/bin/bash
# r04 session 33: HEAD of the round: full GPU suite + smoke() + one more randomized stress seed
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s33
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r04s33/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r04s33/pytest_gpu.log
grep -n "passed\|failed" gpurun_out/r04s33/pytest_gpu.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r04s33/smoke.log
timeout 200 python tests/tools_fuzz_detect.py 45 5 2>&1 | tail -1 | tee gpurun_out/r04s33/fuzz.log
