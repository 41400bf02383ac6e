python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or one_megabase or fresh or full_size_properties" > gpurun_out/r3t_tests.log 2>&1 || { tail -30 gpurun_out/r3t_tests.log; exit 1; }
python bench.py --no-cpu-baseline --no-from-host --steps 10 > gpurun_out/r3t_bench.json 2> gpurun_out/r3t_bench.err || exit 1
REF_N=1000000 READS=4000000 MODE=rmi ITERS=4 bash tools/pmc_quick.sh "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_READ_sum" > gpurun_out/r3t_pmc.txt 2>&1
