# A/B of two builds of the library in one session, alternating: `ab_lib.sh LIB_A LIB_B [rounds]` -> frames/s of the default two-lane bench
R=$GRAFT_REPO_ROOT
A=$1; B=$2; N=${3:-3}
for i in $(seq 1 $N); do
  for L in $A $B; do
    WTK_HIP_LIB=$R/$L timeout -k 10 200 python3 $R/bench.py --no-fp32 --cpu-frames 0 --no-profile > $R/gpurun_out/ab_tmp.log 2>&1 || { echo "bench failed for $L"; tail -5 $R/gpurun_out/ab_tmp.log; exit 1; }
    python3 -c "
import json,sys
j=json.loads(open('$R/gpurun_out/ab_tmp.log').read().strip().splitlines()[-1]); print('$L', round(j['value']), 'frames/s  median window', round(j['windows']['median_ms'],2), 'ms')"
  done
done
