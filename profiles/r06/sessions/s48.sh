# final tree: randomised stress after the last kernel change (k_hog's LDS layout): two seeds of the detector fuzz, one of the split-bank fuzz
mkdir -p gpurun_out/r06_s48
python tests/tools_fuzz_detect.py 180 81 > gpurun_out/r06_s48/fuzz_detect_seed81.log 2>&1; tail -1 gpurun_out/r06_s48/fuzz_detect_seed81.log
python tests/tools_fuzz_detect.py 180 82 > gpurun_out/r06_s48/fuzz_detect_seed82.log 2>&1; tail -1 gpurun_out/r06_s48/fuzz_detect_seed82.log
python tests/tools_fuzz_split.py 60 83 > gpurun_out/r06_s48/fuzz_split_seed83.log 2>&1; tail -1 gpurun_out/r06_s48/fuzz_split_seed83.log
