# round-6 evidence set (final kernels): bench lines, batch-16 and single-frame kernel traces, HBM traffic passes, SQ counter passes
bash profiles/collect_r05.sh r06 bench trace8 traceseq pmc8 sq sq2 sq3 f64
python bench.py --width 1920 --height 1080 --steps 10 --warmup 2 > gpurun_out/r06/bench_1080p.json 2>> gpurun_out/r06/bench_n1.err
ls -la gpurun_out/r06 | head -40
