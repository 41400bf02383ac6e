# read-length sweep of the whole call on the 100 kb reference (off-config points; bench lines go to gpurun_out/sweep/)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sweep; mkdir -p $O
for L in 100 150 250 500 1000 2000 4000 8000; do
  n=$((150000000 / L))
  timeout -k 10 300 python $R/bench.py --read-len $L --reads $n --steps 10 --warmup 2 --no-cpu-baseline --no-from-host --no-other-configs > $O/len_$L.json 2> $O/len_$L.err || exit 1
done
for k in random; do
  timeout -k 10 300 python $R/bench.py --read-kind $k --steps 10 --warmup 2 --no-cpu-baseline --no-from-host --no-other-configs > $O/kind_$k.json 2> $O/kind_$k.err || exit 1
  timeout -k 10 300 python $R/bench.py --read-kind $k --read-len 2000 --reads 75000 --steps 10 --warmup 2 --no-cpu-baseline --no-from-host --no-other-configs > $O/kind_${k}_2000.json 2> $O/kind_${k}_2000.err || exit 1
done
python3 - <<PY
import json, glob, os
rows = []
for f in sorted(glob.glob("$O/*.json")):
    j = json.load(open(f)); c = j["config"]
    rows.append({"file": os.path.basename(f), "read_len": c["read_len"], "reads": c["reads_per_gpu_per_step"], "kind": c["read_distribution"],
                 "G_bases_per_s": round(j["value"] / 1e9, 2), "ms_per_step": round(j["ms_per_step"], 3), "search_kernel_ms": round(j["roofline"]["kernel_ms_avg"], 3),
                 "smems_per_read": c["smems_per_read"]})
json.dump(rows, open("$O/summary.json", "w"), indent=1)
for r in rows: print(r)
PY
