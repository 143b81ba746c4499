# A/B of environment settings under the hybrid bench (`bench.py --dtype hybrid`), round-robin: `ab_env_hyb.sh ROUNDS "K=V" ...`
R=$GRAFT_REPO_ROOT
N=$1; shift
cd $R
for i in $(seq 1 $N); do
  for kv in "$@"; do
    env $kv timeout -k 10 200 python3 bench.py --dtype hybrid --cpu-frames 0 --no-profile > gpurun_out/abenv_tmp.log 2>&1 || { echo "failed $kv"; tail -3 gpurun_out/abenv_tmp.log; continue; }
    python3 -c "
import json
j=json.loads(open('gpurun_out/abenv_tmp.log').read().strip().splitlines()[-1]); print('%-28s %6.0f frames/s  median window %.2f ms' % ('$kv', j['value'], j['windows']['median_ms']))"
  done
done
