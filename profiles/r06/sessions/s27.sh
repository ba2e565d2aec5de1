# zero-copy argmin tree: the whole GPU suite + smoke; then the probe build's per-phase block trace (single frame, batch of 8) for the fold and plain launches
mkdir -p gpurun_out/r06_s27
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06_s27/pytest_gpu.log 2>&1
tail -3 gpurun_out/r06_s27/pytest_gpu.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_s27/smoke.log 2>&1; tail -1 gpurun_out/r06_s27/smoke.log
for l in 0 1 2 3; do python tests/tools_dt_trace.py 640 480 $l > gpurun_out/r06_s27/trace_single_l$l.txt 2>&1; done
for l in 0 1 2 3; do python tests/tools_dt_trace.py 640 480 $l 8 > gpurun_out/r06_s27/trace_b8_l$l.txt 2>&1; done
grep -h "batch of 8" gpurun_out/r06_s27/trace_b8_l*.txt
