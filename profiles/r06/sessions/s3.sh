bash profiles/r06/sessions/ab.sh r06_s3 3 libpbd_hip_base.so libpbd_hip.so
for l in 0 1 2; do python tests/tools_dt_trace.py 640 480 $l 16 2>&1 | grep "batch of"; done
