export REF_N=1000000 READS=4000000 MODE=rmi ITERS=4
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "compact or randomized or one_megabase or sharded or config4" > gpurun_out/r3d_tests.log 2>&1 || { tail -30 gpurun_out/r3d_tests.log; exit 1; }
python bench.py --config 3 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r3d_cfg3.json 2> gpurun_out/r3d_cfg3.err || exit 1
bash tools/pmc_quick.sh "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_READ_sum" > gpurun_out/r3d_pmc.txt 2>&1 || exit 1
RUNS="SEARCH_ONLY=1
SEARCH_ONLY=1 BPC=3
SEARCH_ONLY=1 BPC=2
SEARCH_ONLY=1 BPC=1" bash tools/ka_sweep.sh > gpurun_out/r3d_sweep.txt 2>&1
