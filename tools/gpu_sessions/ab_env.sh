# A/B an environment switch in one GPU session: usage ab_env.sh VAR valA valB
R=$GRAFT_REPO_ROOT
VAR=$1; A=$2; B=$3
for rep in 1 2 3; do
  for val in $A $B; do
    env $VAR=$val WTK_NO_SIDE_STREAM=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --lanes 1 > gpurun_out/ab_env.log 2>&1
    echo $rep $VAR=$val $(python -c "import json; d=json.loads(open('gpurun_out/ab_env.log').read().strip().splitlines()[-1]); print(round(d['value']), round(d['roofline']['class_ms_per_step']['conv'],4))")
  done
done
cd /tmp && export TMPDIR=/tmp WTK_NO_SIDE_STREAM=1
for val in $A $B; do
  env $VAR=$val timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_ab_$val -o bench -- python3 $R/bench.py --steps 8 --warmup 3 --cpu-frames 0 --lanes 1 --no-profile > $R/gpurun_out/prof_ab_$val.log 2>&1 || echo fail $val
done
