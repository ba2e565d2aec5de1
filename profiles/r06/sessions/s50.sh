# final tree: a longer randomised stress (two more seeds of the detector fuzz, one of the split-bank fuzz, the double-precision detector fuzz if the tool takes a dtype)
mkdir -p gpurun_out/r06_s50
python tests/tools_fuzz_detect.py 300 91 > gpurun_out/r06_s50/fuzz_detect_seed91.log 2>&1; tail -1 gpurun_out/r06_s50/fuzz_detect_seed91.log
python tests/tools_fuzz_detect.py 300 92 > gpurun_out/r06_s50/fuzz_detect_seed92.log 2>&1; tail -1 gpurun_out/r06_s50/fuzz_detect_seed92.log
python tests/tools_fuzz_split.py 90 93 > gpurun_out/r06_s50/fuzz_split_seed93.log 2>&1; tail -1 gpurun_out/r06_s50/fuzz_split_seed93.log
