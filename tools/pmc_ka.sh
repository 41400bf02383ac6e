# PMC passes for the search kernel (each --pmc set in its own run, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_ka; mkdir -p $O
rocprofv3 -L > $O/counters_list.txt 2>&1 || true
i=0
for set in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_REQ_sum TCC_IB_STALL_sum" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_SERIALIZATION_STALL_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/p$i.log 2>&1 || echo "pass $i failed" >> $O/failed.txt
done
python3 - <<'PY'
import csv,glob,collections,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/pmc_ka"
for f in sorted(glob.glob(O+"/p*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "match_table" in k or "match_stats" in k: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c,v in agg.items(): print(c, sum(v)/len(v), len(v))
PY
