#!/bin/bash
# r05 session 28: last look at the committed tree — the round's new paths (image depths, tuner, binary16 bank, benched unit at 16 frames) and smoke
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s28; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x -k "wide or detect_image or tune_plan or split_f16 or benched_unit or smoke" > $O/pytest_new.log 2>&1; echo "rc=$?" >> $O/pytest_new.log; tail -3 $O/pytest_new.log
timeout 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
