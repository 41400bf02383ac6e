python -m pytest tests -m gpu -x -q > gpurun_out/r3v_tests.log 2>&1 || { tail -30 gpurun_out/r3v_tests.log; exit 1; }
python bench.py --no-cpu-baseline --steps 10 > gpurun_out/r3v_bench.json 2> gpurun_out/r3v_bench.err || exit 1
RUNS="ITERS=8" bash tools/ka_sweep.sh > gpurun_out/r3v_sweep.txt 2>&1
