python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r3i_tests.log 2>&1 || { tail -40 gpurun_out/r3i_tests.log; exit 1; }
python bench.py --no-cpu-baseline --no-other-configs --steps 10 > gpurun_out/r3i_bench.json 2> gpurun_out/r3i_bench.err || exit 1
GENIE_BENCH_NBUF=4 python bench.py --no-cpu-baseline --no-other-configs --steps 10 > gpurun_out/r3i_nbuf4.json 2> gpurun_out/r3i_nbuf4.err || exit 1
python bench.py --config 5 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r3i_cfg5.json 2> gpurun_out/r3i_cfg5.err || exit 1
