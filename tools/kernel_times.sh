# per-kernel average times for a few batch sizes (kernel trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in $SIZES; do
  O=$R/gpurun_out/kt_$n; mkdir -p $O
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O -o kt --output-format csv -- python3 $R/bench.py --reads $n --steps 10 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/err.txt || exit 1
  python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$O/kt_kernel_stats.csv")) if "genie" in r["Name"]]
print("reads $n:", "; ".join("%s %.1f us"%(r["Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0].split("::")[-1][:28], float(r["AverageNs"])/1e3) for r in rows[:3]))
PY
done
