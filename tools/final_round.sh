# Round-end measurements: default bench (with the CPU baseline), the other modes, config 3, then the profile passes.
TAG=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final_$TAG; mkdir -p $O
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
for m in bwa rmi; do timeout -k 10 200 python bench.py --mode $m --no-cpu-baseline > $O/bench_$m.json 2> $O/bench_$m.err || exit 1; done
timeout -k 10 500 python bench.py --config 3 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err || exit 1
bash tools/profile_round.sh $TAG || exit 1
python - <<PY
import json
for n in ("default","bwa","rmi","cfg3"):
    j=json.load(open("$O/bench_%s.json"%n)); r=j["roofline"]
    print(n, "%.2f G"%(j["value"]/1e9), "%.3f ms/step"%j["ms_per_step"], "K_A %.3f ms"%r["kernel_ms_avg"], "frac %.2f"%r["frac"], j.get("cpu_baseline",{}).get("value"))
PY
