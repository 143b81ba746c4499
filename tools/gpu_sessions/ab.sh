# A/B two builds of the library in one GPU session: alternating runs, single stream, per-class device time
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for lib in base new; do
    if [ $lib = base ]; then export WTK_HIP_LIB=$R/wtracker_amd/libwtk_hip_base.so; else unset WTK_HIP_LIB; fi
    WTK_NO_SIDE_STREAM=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --lanes 1 > gpurun_out/ab_$lib.log 2>&1
    echo $rep $lib $(python -c "import json; d=json.loads(open('gpurun_out/ab_$lib.log').read().strip().splitlines()[-1]); print(round(d['value']), round(d['roofline']['class_ms_per_step']['conv'],4))")
  done
done
