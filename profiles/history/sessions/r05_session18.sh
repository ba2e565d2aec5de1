#!/bin/bash
# r05 session 18: pbd_detect_image — 16-bit / float / double images (pyramid, HOG, detect) against the oracle; the u8 pyramid / HOG tests beside them
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s18; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "wide or detect_image or resize_bit_exact or pyrdown_bit_exact or hog_bit_exact or pyramid_levels" > $O/pytest_depths.log 2>&1; echo "rc=$?" >> $O/pytest_depths.log; tail -25 $O/pytest_depths.log
