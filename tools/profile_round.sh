# Round profile of one bench workload: kernel trace + stats, then every counter set in its OWN --pmc pass
# (FETCH_SIZE and WRITE_SIZE apart, as MI355X_MICROARCH.md section HBM prescribes; kernel-trace only).
# usage (on the GPU box): bash tools/profile_round.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-from-host --no-other-configs"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O -o kt --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 $B "$@" > $O/kt_bench.json 2> $O/kt.err || { tail -3 $O/kt.err; exit 1; }
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_GATE_EN1_sum TCC_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_READ_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 $B "$@" > $O/p$i.json 2> $O/p$i.err || { echo "pass $i ($set) failed" >> $O/failed.txt; tail -3 $O/p$i.err; }
done
python3 $R/profiles/summarize.py $O $TAG
