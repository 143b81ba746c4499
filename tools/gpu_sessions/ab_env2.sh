# A/B of an environment switch under the DEFAULT bench (two lanes, side stream): frames/s is the metric here, because
# single-stream kernel time cannot see work that only fills CUs the other forward pass would have used
# usage: bash tools/gpu_sessions/ab_env2.sh VAR a b [reps]
VAR=$1; A=$2; B=$3; REPS=${4:-3}
for rep in $(seq $REPS); do
  for val in $A $B; do
    env $VAR=$val timeout -k 10 200 python bench.py --steps 40 --warmup 5 --cpu-frames 0 --no-profile > gpurun_out/ab_env2.log 2>&1 || exit 1
    echo $rep $VAR=$val $(python -c "import json; d=json.loads(open('gpurun_out/ab_env2.log').read().strip().splitlines()[-1]); print(round(d['value']))")
  done
done
