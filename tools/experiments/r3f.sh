python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r3f_tests.log 2>&1 || { tail -30 gpurun_out/r3f_tests.log; exit 1; }
python bench.py > gpurun_out/r3f_bench.json 2> gpurun_out/r3f_bench.err || exit 1
RUNS="SEARCH_ONLY=1
SEARCH_ONLY=1 FMT=compact
FMT=wide
FMT=compact" bash tools/ka_sweep.sh > gpurun_out/r3f_sweep.txt 2>&1
