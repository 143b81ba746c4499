# per-layer single-stream traces of two library builds (base = wtracker_amd/libwtk_hip_base.so through WTK_HIP_LIB)
cd /tmp && export TMPDIR=/tmp WTK_NO_SIDE_STREAM=1
R=$GRAFT_REPO_ROOT
for lib in base new; do
  if [ $lib = base ]; then export WTK_HIP_LIB=$R/wtracker_amd/libwtk_hip_base.so; else unset WTK_HIP_LIB; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_lib_$lib -o bench -- python3 $R/bench.py --steps 8 --warmup 3 --cpu-frames 0 --lanes 1 --no-profile > $R/gpurun_out/prof_lib_$lib.log 2>&1 || echo fail $lib
done
echo done
