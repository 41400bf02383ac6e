export REF_N=1000000 READS=4000000 MODE=rmi ITERS=4
bash tools/pmc_quick.sh "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" > gpurun_out/r3c_p9a.txt 2>&1 &&
bash tools/pmc_quick.sh "TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum TCC_READ_sum TCC_TAG_STALL_sum" > gpurun_out/r3c_p9b.txt 2>&1 &&
TABLE_BITS=10 bash tools/pmc_quick.sh "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" > gpurun_out/r3c_p10a.txt 2>&1 &&
RUNS="SEARCH_ONLY=1
SEARCH_ONLY=1 BPC=1
SEARCH_ONLY=1 TABLE_BITS=10
SEARCH_ONLY=1 TABLE_BITS=8" bash tools/ka_sweep.sh > gpurun_out/r3c_sweep.txt 2>&1
