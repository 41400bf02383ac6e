mkdir -p gpurun_out/cfg3
for v in $AB_DIR2; do
  GENIE_DIR2_BITS=$v timeout -k 10 400 python bench.py --config 3 --reads 2000000 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/cfg3/rmi_p2_$v.json 2> gpurun_out/cfg3/rmi_p2_$v.err || exit 1
  GENIE_DIR2_BITS=$v timeout -k 10 400 python bench.py --config 3 --mode lut --reads 2000000 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/cfg3/lut_p2_$v.json 2> gpurun_out/cfg3/lut_p2_$v.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/cfg3/*.json")):
    j=json.load(open(f)); r=j["roofline"]
    print(f.split("/")[-1], "%.2f G"%(j["value"]/1e9), "%.3f ms/step"%j["ms_per_step"], "K_A %.3f ms"%r["kernel_ms_avg"], "path %.3f"%r["path"]["ms_avg"], j["config"]["index_build_plus_broadcast_s"])
PY
