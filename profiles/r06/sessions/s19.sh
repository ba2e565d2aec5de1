bash profiles/r06/sessions/ab.sh r06_s19 3 libpbd_hip.so libpbd_hip_kp.so
