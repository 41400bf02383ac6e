# experiment: K_A time with stages switched off (results invalid; timing only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in $DBGS; do
  O=$R/gpurun_out/dbg_$d; mkdir -p $O
  GENIE_DBG=$d timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O -o kt --output-format csv -- python3 $R/tools/ka_only.py > $O/out.txt 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
  python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$O/kt_kernel_stats.csv")) if "genie" in r["Name"]]
print("dbg $d:", "; ".join("%s %.1f us"%(r["Name"].split("::")[-1][:22], float(r["AverageNs"])/1e3) for r in rows[:3]))
PY
done
