#!/bin/bash
# session 37: final collection of the round (r03d: after the scan-loop trims)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
SKIP_CONV_MODES=1 bash profiles/collect.sh r03d > gpurun_out/collect_r03d.log 2>&1
cat gpurun_out/r03d/sweep_sb.txt gpurun_out/r03d/batch_stages.txt
tail -2 gpurun_out/collect_r03d.log
