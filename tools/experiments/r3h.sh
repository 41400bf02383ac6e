python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r3h_tests.log 2>&1 || { tail -30 gpurun_out/r3h_tests.log; exit 1; }
python bench.py > gpurun_out/r3h_bench.json 2> gpurun_out/r3h_bench.err || exit 1
for nb in 3 4; do GENIE_BENCH_NBUF=$nb python bench.py --no-cpu-baseline --no-other-configs --steps 5 > gpurun_out/r3h_nbuf$nb.json 2> gpurun_out/r3h_nbuf$nb.err || exit 1; done
RUNS="SEARCH_ONLY=1
SEARCH_ONLY=1 GROUP_POS=1200
SEARCH_ONLY=1 GROUP_POS=1500
SEARCH_ONLY=1 GROUP_POS=600" bash tools/ka_sweep.sh > gpurun_out/r3h_sweep.txt 2>&1
