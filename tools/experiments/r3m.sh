python -m pytest tests -m gpu -x -q > gpurun_out/r3m_tests.log 2>&1 || { tail -40 gpurun_out/r3m_tests.log; exit 1; }
python bench.py --no-cpu-baseline --steps 10 > gpurun_out/r3m_bench.json 2> gpurun_out/r3m_bench.err || exit 1
bash tools/pmc_quick.sh "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum SQ_INSTS_VALU" > gpurun_out/r3m_pmc.txt 2>&1
