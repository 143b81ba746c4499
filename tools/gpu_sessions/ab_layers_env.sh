# per-layer single-stream traces for two values of an environment switch: bash tools/gpu_sessions/ab_layers_env.sh VAR a b
cd /tmp && export TMPDIR=/tmp WTK_NO_SIDE_STREAM=1
R=$GRAFT_REPO_ROOT
VAR=$1
for val in $2 $3; do
  export $VAR=$val
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_env_$val -o bench -- python3 $R/bench.py --steps 8 --warmup 3 --cpu-frames 0 --lanes 1 --no-profile > $R/gpurun_out/prof_env_$val.log 2>&1 || echo fail $val
done
echo done
