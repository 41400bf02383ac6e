# Round-end records: bench lines of every config (+ BWA mode, + the off-BASELINE 2.5 Mb reference), the read-length /
# read-kind sweep, then the profile passes of configs 1-3 (summarized afterwards with profiles/summarize.py from the merged
# gpurun_out/prof_<tag>_cfg*/).  usage (on the GPU box): bash tools/final_round.sh <tag>  -> gpurun_out/final_<tag>/
# A third argument picks a part (one gpurun call is at most 20 minutes): bench | sweep | prof1 | prof2 | prof3; none = everything.
TAG=$1; PART=${2:-all}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final_$TAG; mkdir -p $O
if [ $PART = all ] || [ $PART = bench ]; then
timeout -k 10 600 python bench.py > $O/bench_cfg1.json 2> $O/bench_cfg1.err || exit 1
timeout -k 10 500 python bench.py --config 2 --no-other-configs > $O/bench_cfg2.json 2> $O/bench_cfg2.err || exit 1
timeout -k 10 300 python bench.py --mode bwa --no-cpu-baseline --no-from-host --no-other-configs > $O/bench_bwa.json 2> $O/bench_bwa.err || exit 1
timeout -k 10 600 python bench.py --config 3 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err || exit 1
timeout -k 10 900 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err || exit 1
timeout -k 10 600 python bench.py --config 5 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err || exit 1
fi
if [ $PART = all ] || [ $PART = sweep ]; then
bash tools/sweep_read_len.sh > $O/sweep.txt 2>&1 || exit 1
cp $R/gpurun_out/sweep/summary.json $O/read_len_sweep.json
fi
for c in 1 2 3; do
  if [ $PART = all ] || [ $PART = prof$c ]; then bash tools/profile_round.sh ${TAG}_cfg$c --config $c > $O/prof_cfg$c.txt 2>&1 || exit 1; fi
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    j = json.load(open(f)); r = j["roofline"]
    print(f.split("/")[-1], "%.1f G" % (j["value"] / 1e9), "%.3f ms" % j["ms_per_step"], "K_A %.3f" % r["kernel_ms_avg"], "frac %.3f" % r["frac"],
          (j.get("value_from_host") or {}).get("value"), (j.get("cpu_baseline") or {}).get("value"))
PY
