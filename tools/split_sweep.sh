mkdir -p gpurun_out/split
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/split/tests.log 2>&1 || { tail -20 gpurun_out/split/tests.log; exit 1; }
tail -1 gpurun_out/split/tests.log
for m in $MODES; do for v in $SPLITS; do
  GENIE_PAIR_SPLIT=$v timeout -k 10 120 python bench.py --mode $m --search-all --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/split/${m}_$v.json 2> gpurun_out/split/${m}_$v.err || exit 1
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/split/*.json")):
    j=json.load(open(f)); r=j["roofline"]
    print(f.split("/")[-1], "%.2f G"%(j["value"]/1e9), "%.3f ms/step"%j["ms_per_step"], "K_A %.3f ms"%r["kernel_ms_avg"])
PY
