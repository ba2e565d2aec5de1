mkdir -p gpurun_out/r06_s12
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06_s12/pytest_gpu.log 2>&1
tail -3 gpurun_out/r06_s12/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_s12/smoke.log 2>&1; tail -1 gpurun_out/r06_s12/smoke.log
