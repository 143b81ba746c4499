# One-switch-at-a-time sweep of the A/B environment switches under the default two-lane bench: `env_sweep.sh [rounds]`
R=$GRAFT_REPO_ROOT
N=${1:-2}
cd $R
for i in $(seq 1 $N); do
  for kv in DEFAULT=1 WTK_HALO_PERSIST=0 WTK_HALO_SMALL_BLOCKS=0 WTK_NO_WIDE_1X1=1 WTK_NO_WS64=1 WTK_NO_S2WIN=1 WTK_NO_IGEMM_TAIL=1 WTK_NO_FUSED_TAIL=1 WTK_NO_SIDE_STREAM=1 WTK_MATERIALIZE_UPSAMPLE=1; do
    env $kv timeout -k 10 200 python3 bench.py --no-fp32 --cpu-frames 0 --no-profile > gpurun_out/sweep_tmp.log 2>&1 || { echo "failed $kv"; continue; }
    python3 -c "
import json
j=json.loads(open('gpurun_out/sweep_tmp.log').read().strip().splitlines()[-1]); print('%-28s %6.0f frames/s  median window %.2f ms' % ('$kv', j['value'], j['windows']['median_ms']))"
  done
done
