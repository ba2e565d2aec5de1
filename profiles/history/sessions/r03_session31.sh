#!/bin/bash
# session 31: overlap of the filter bank and the DP when the filter bank leaves LDS to the DT blocks (tuning build)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03ac; mkdir -p $OUT; cd $REPO
export PBD_LIBRARY=$REPO/partsbaseddetector_amd/libpbd_hip_tune.so
for kb in 0 41 54 81; do
  echo "== PBD_CONV_LDS_KB=$kb" >> $OUT/summary.txt
  PBD_CONV_LDS_KB=$kb python tests/tools_overlap_probe.py 0.3 2>/dev/null | grep -E "^(2 pdf \+ 0|0 pdf \+ 3|2 pdf \+ 2|3 pdf \+ 3|2 pdf \+ 4)" >> $OUT/summary.txt
done
cat $OUT/summary.txt
