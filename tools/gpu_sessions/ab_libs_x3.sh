# A/B of several builds of the library on the f16x3 mode (the headline), round-robin: `ab_libs_x3.sh ROUNDS LIB...` -> frames/s, two lanes and one
R=$GRAFT_REPO_ROOT
N=$1; shift
for i in $(seq 1 $N); do
  for L in "$@"; do
    for lanes in 2 1; do
      WTK_HIP_LIB=$R/$L timeout -k 10 200 python3 $R/bench.py --dtype f16x3 --no-fp32 --cpu-frames 0 --no-profile --repeats 5 --lanes $lanes > $R/gpurun_out/ab_tmp.log 2>&1 || { echo "bench failed for $L"; tail -5 $R/gpurun_out/ab_tmp.log; exit 1; }
      python3 -c "
import json
j=json.loads(open('$R/gpurun_out/ab_tmp.log').read().strip().splitlines()[-1]); print('$L lanes $lanes', round(j['value']), 'frames/s  median window', round(j['windows']['median_ms'],2), 'ms')"
    done
  done
done
