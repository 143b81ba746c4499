# A/B of several builds of the library in one session, round-robin: `ab_libs.sh ROUNDS LIB...` -> frames/s of the default two-lane bench
R=$GRAFT_REPO_ROOT
N=$1; shift
for i in $(seq 1 $N); do
  for L in "$@"; do
    WTK_HIP_LIB=$R/$L timeout -k 10 200 python3 $R/bench.py --no-fp32 --cpu-frames 0 --no-profile > $R/gpurun_out/ab_tmp.log 2>&1 || { echo "bench failed for $L"; tail -5 $R/gpurun_out/ab_tmp.log; exit 1; }
    python3 -c "
import json
j=json.loads(open('$R/gpurun_out/ab_tmp.log').read().strip().splitlines()[-1]); print('$L', round(j['value']), 'frames/s  median window', round(j['windows']['median_ms'],2), 'ms')"
  done
done
