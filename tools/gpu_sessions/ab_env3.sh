# A/B of one environment switch under the default bench (frames/s, two lanes) AND single-stream per-kernel time: `ab_env3.sh VAR a b [reps]`
R=$GRAFT_REPO_ROOT
VAR=$1; A=$2; B=$3; REPS=${4:-3}
cd $R
for i in $(seq 1 $REPS); do
  for v in $A $B; do
    env $VAR=$v python bench.py --cpu-frames 0 --no-fp32 --repeats 5 > gpurun_out/ab3_${VAR}_${v}_$i.json 2> gpurun_out/ab3_${VAR}_${v}_$i.err || echo "run failed $v $i"
    python - <<PY
import json
j=json.load(open("gpurun_out/ab3_${VAR}_${v}_$i.json"))
r=j["roofline"]
print("${VAR}=${v} run $i: %.0f frames/s  windows %.2f..%.2f ms  | "%(j["value"], j["windows"]["min_ms"], j["windows"]["max_ms"]) + "  ".join("%s %.0f TF/s %.3f ms"%(k["kernel"].split("_kernel")[0][-12:], k["achieved"], k["ms_per_step"]) for k in r["kernels"]))
PY
  done
done
