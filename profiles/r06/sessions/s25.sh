bash profiles/r06/sessions/ab.sh r06_s25 2 libpbd_hip.so libpbd_hip_st48.so libpbd_hip_st100.so
