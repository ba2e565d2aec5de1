mkdir -p gpurun_out/r06_s22
python tests/tools_fuzz_detect.py 240 61 > gpurun_out/r06_s22/fuzz_detect_seed61.log 2>&1; tail -2 gpurun_out/r06_s22/fuzz_detect_seed61.log
python tests/tools_fuzz_detect.py 240 62 > gpurun_out/r06_s22/fuzz_detect_seed62.log 2>&1; tail -2 gpurun_out/r06_s22/fuzz_detect_seed62.log
python tests/tools_fuzz_split.py 60 63 > gpurun_out/r06_s22/fuzz_split_seed63.log 2>&1; tail -1 gpurun_out/r06_s22/fuzz_split_seed63.log
