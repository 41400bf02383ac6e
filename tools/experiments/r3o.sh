python -m pytest tests -m gpu -x -q > gpurun_out/r3o_tests.log 2>&1 || { tail -40 gpurun_out/r3o_tests.log; exit 1; }
python bench.py --no-cpu-baseline --no-from-host --steps 10 > gpurun_out/r3o_bench.json 2> gpurun_out/r3o_bench.err || exit 1
bash tools/pmc_quick.sh "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum SQ_INSTS_VALU" > gpurun_out/r3o_pmc.txt 2>&1
