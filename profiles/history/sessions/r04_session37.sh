#!/bin/bash
# r04 session 37 (run twice: first kernel, then with the D[filter][cell] / n-tile variants): the last GPU minute of the round: the bf16 split-product filter bank probe (tests/tools/conv_split_probe.hip, built here
# with hipcc, not part of the product): does the arithmetic hold on the hardware, and what rate does a first kernel reach?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s37
timeout 60 tests/tools/conv_split_probe 640 512 2>&1 | tee gpurun_out/r04s37/probe6.log
