#!/bin/bash
# r04 session 35: last GPU minutes of the round: one more randomized stress seed on HEAD
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s35
timeout 150 python tests/tools_fuzz_detect.py 70 6 2>&1 | tail -1 | tee gpurun_out/r04s35/fuzz.log
