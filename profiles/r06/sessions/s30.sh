# task-descriptor prefetch 64 / 256 / 1024 blocks ahead (vector load into this XCD's L2) against the product build (DT priority 1, merged z table, fused sums); CU spread of a single frame's launches
mkdir -p gpurun_out/r06_s30
bash profiles/r06/sessions/ab.sh r06_s30 3 libpbd_hip.so libpbd_hip_tp64.so libpbd_hip_tp256.so libpbd_hip_tp1024.so
python tests/tools_dt_cu_spread.py 2>&1 | grep launch > gpurun_out/r06_s30/cu_spread.txt
