mkdir -p gpurun_out/cfg3
for v in $AB_DIR2; do for m in $MODES; do
  GENIE_DIR2_BITS=$v timeout -k 10 500 python bench.py --config 3 --mode $m --reads $READS --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/cfg3/${m}_p2_$v.json 2> gpurun_out/cfg3/${m}_p2_$v.err || exit 1
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/cfg3/*.json")):
    j=json.load(open(f)); r=j["roofline"]
    print(f.split("/")[-1], "%.2f G"%(j["value"]/1e9), "%.3f ms/step"%j["ms_per_step"], "K_A %.3f ms"%r["kernel_ms_avg"], "path %.3f"%r["path"]["ms_avg"], j["config"]["index_build_plus_broadcast_s"])
PY
