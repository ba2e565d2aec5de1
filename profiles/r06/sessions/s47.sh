# final tree: the probe build's per-phase block traces once more (single frame and batch of 8; leaf x, y, fold x, y launches) — the "after" of profiles/r06/dt_phase_trace/r06_session27_*
mkdir -p gpurun_out/r06_s47
for l in 0 1 2 3; do python tests/tools_dt_trace.py 640 480 $l > gpurun_out/r06_s47/trace_single_l$l.txt 2>&1; done
for l in 0 1 2 3; do python tests/tools_dt_trace.py 640 480 $l 8 > gpurun_out/r06_s47/trace_b8_l$l.txt 2>&1; done
grep -h "batch of 8" gpurun_out/r06_s47/trace_b8_l*.txt
python tests/tools_hog_probe.py 2>&1 | grep phases | tail -1
