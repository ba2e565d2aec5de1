# final tree (commit 0afe388): the whole GPU suite, smoke, the round-6 evidence set once more
mkdir -p gpurun_out/r06_s45
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06_s45/pytest_gpu.log 2>&1
grep -E "passed|failed" gpurun_out/r06_s45/pytest_gpu.log | tail -1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_s45/smoke.log 2>&1; tail -1 gpurun_out/r06_s45/smoke.log
rm -rf gpurun_out/r06
bash profiles/collect_r05.sh r06 bench trace8 traceseq pmc8 sq sq2 sq3 f64
python bench.py --width 1920 --height 1080 --steps 10 --warmup 2 > gpurun_out/r06/bench_1080p.json 2>> gpurun_out/r06/bench_n1.err
