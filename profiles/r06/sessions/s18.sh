bash profiles/r06/sessions/ab.sh r06_s18 3 libpbd_hip.so libpbd_hip_pf512.so libpbd_hip_pf1024.so
