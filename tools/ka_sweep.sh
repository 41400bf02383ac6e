# K_A experiments: each line of $RUNS is "ENV=VAL ENV=VAL ..." for tools/ka_only.py; prints the top kernels' average times
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
echo "$RUNS" | while IFS= read -r envs; do
  [ -z "$envs" ] && continue
  i=$((i+1)); O=$R/gpurun_out/kas_$i; mkdir -p $O
  env $envs timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o kt --output-format csv -- python3 $R/tools/ka_only.py > $O/out.txt 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
  python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$O/kt_kernel_stats.csv")) if "genie" in r["Name"]]
print("$envs:", "; ".join("%s %.1f us"%(r["Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0].split("::")[-1][:28], float(r["AverageNs"])/1e3) for r in rows[:4]))
PY
done
