#!/bin/bash
# session 28: the round's second rocprofv3 collection (after the 16-byte B loads of the filter bank)
bash profiles/collect.sh r03b > gpurun_out/collect_r03b.log 2>&1
cat gpurun_out/r03b/sweep_sb.txt gpurun_out/r03b/batch_stages.txt
tail -3 gpurun_out/collect_r03b.log
