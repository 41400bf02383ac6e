mkdir -p gpurun_out/sweep
for spec in "150 1000000" "250 600000" "500 300000" "1000 150000" "2000 80000" "5000 30000" "8000 20000"; do
  set -- $spec
  for m in lut bwa rmi; do
    timeout -k 10 120 python bench.py --mode $m --read-len $1 --reads $2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/sweep/L$1_$m.json 2> gpurun_out/sweep/L$1_$m.err || exit 1
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/sweep/*.json"), key=lambda f:(int(f.split("/L")[1].split("_")[0]), f)):
    j=json.load(open(f)); print(f.split("/")[-1], "%.2f G"%(j["value"]/1e9), "%.3f ms"%j["ms_per_step"], j["config"]["smems_per_read"])
PY
