RUNS="SEARCH_ONLY=1
SEARCH_ONLY=1 DBG=64
SEARCH_ONLY=1 DBG=128
SEARCH_ONLY=1 DBG=256
SEARCH_ONLY=1 DBG=448" bash tools/ka_sweep.sh > gpurun_out/r3u_sweep.txt 2>&1
REF_N=1000000 READS=4000000 MODE=rmi ITERS=4 RUNS="SEARCH_ONLY=1
SEARCH_ONLY=1 DBG=64
SEARCH_ONLY=1 DBG=128
SEARCH_ONLY=1 DBG=256
SEARCH_ONLY=1 DBG=448" bash tools/ka_sweep.sh >> gpurun_out/r3u_sweep.txt 2>&1
for d in 64 128 256; do echo "dbg $d"; REF_N=1000000 READS=4000000 MODE=rmi ITERS=4 SEARCH_ONLY=1 DBG=$d bash tools/pmc_quick.sh "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_READ_sum" 2>&1 | grep match_table; done >> gpurun_out/r3u_sweep.txt
