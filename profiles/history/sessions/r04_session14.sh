#!/bin/bash
# r04 session 14: randomized stress of the product path against the oracle (tests/tools_fuzz_detect.py): random trees / mixtures / filter
# sizes / cell sizes / image sizes / gray + colour / float + double / fold, three-kernel and compact plans / graph replay / batches, and
# random stand-alone distance transforms incl. exact ties and weak curvature
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s14
for s in 1 2; do timeout 400 python tests/tools_fuzz_detect.py 90 $s 2>&1 | tail -3 | tee -a gpurun_out/r04s14/fuzz.log; done
