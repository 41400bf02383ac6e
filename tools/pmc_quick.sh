# one --pmc pass (counters in $1) of tools/ka_only.py; prints per-kernel averages per launch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcq; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $1 -d $O -o p --output-format csv -- python3 $R/tools/ka_only.py > $O/out.txt 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
python3 - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(match_table\w*|traverse_\w*kernel|interval_kernel)", r["Kernel_Name"])
        if m: agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in sorted(d.items())})
PY
