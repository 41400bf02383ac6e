# A/B helper: GPU parity tests, then short bench runs (no CPU baseline) -> gpurun_out/ab/*.json
mkdir -p gpurun_out/ab
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/ab/tests.log 2>&1 || { tail -20 gpurun_out/ab/tests.log; exit 1; }
tail -1 gpurun_out/ab/tests.log
for m in lut bwa rmi; do
  timeout -k 10 120 python bench.py --mode $m --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/ab/$m.json 2> gpurun_out/ab/$m.err || exit 1
done
for v in $AB_DIR2; do
  for m in lut bwa; do
    GENIE_DIR2_BITS=$v timeout -k 10 120 python bench.py --mode $m --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/ab/${m}_p2_$v.json 2> gpurun_out/ab/${m}_p2_$v.err || exit 1
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ab/*.json")):
    j=json.load(open(f)); r=j["roofline"]
    print(f.split("/")[-1], "%.2f G"%(j["value"]/1e9), "%.3f ms/step"%j["ms_per_step"], "K_A %.3f ms"%r["kernel_ms_avg"], "path %.3f"%r["path"]["ms_avg"])
PY
