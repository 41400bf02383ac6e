# SQ instruction-mix and wait counters for every kernel of the pipeline (each --pmc set in its own run)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_all; mkdir -p $O
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $O/p$i.log 2>&1 || echo "pass $i failed" >> $O/failed.txt
done
python3 - <<'PY'
import csv,glob,collections,os,re
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/pmc_all"
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(O+"/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        m=re.search(r"(match_table_kernel|match_stats\w*kernel|traverse_kernel|interval_kernel)", r["Kernel_Name"])
        if m: agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in agg.items():
    print(k)
    for c,v in sorted(d.items()): print("   %-36s %16.0f  (%d launches)"%(c, sum(v)/len(v), len(v)))
PY
