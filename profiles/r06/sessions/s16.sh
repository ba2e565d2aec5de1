bash profiles/r06/sessions/ab.sh r06_s16 5 libpbd_hip_nolive.so libpbd_hip.so
