# A/B of two library builds under the DEFAULT bench (two lanes, side stream): wtracker_amd/libwtk_hip_base.so (through
# WTK_HIP_LIB) against the current build, alternating runs; frames/s is the metric (see ab_env2.sh)
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for lib in base new; do
    if [ $lib = base ]; then export WTK_HIP_LIB=$R/wtracker_amd/libwtk_hip_base.so; else unset WTK_HIP_LIB; fi
    timeout -k 10 200 python bench.py --steps 40 --warmup 5 --cpu-frames 0 --no-profile > gpurun_out/ab2_$lib.log 2>&1 || exit 1
    echo $rep $lib $(python -c "import json; d=json.loads(open('gpurun_out/ab2_$lib.log').read().strip().splitlines()[-1]); print(round(d['value']))")
  done
done
