# per-layer timelines of the latency plan with ONE conv_sk tile forced for every layer (WTK_SK_TILE): which tile wins for which layer shape
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp WTK_NO_SIDE_STREAM=1 WTK_SK_INKERNEL_MAX_KB=1000000
for B in 1 15; do for T in 0 1 2 3; do
  N=tile${T}_b${B}
  WTK_SK_TILE=$T timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trt_$N -o t -- python3 $R/tools/gpu_sessions/time_mode.py --dtype f16x3 --steps 6 --batch $B --size 384 --plan latency > $R/gpurun_out/trt_$N.log 2>&1 || echo "trace $N failed"
  F=$(find $R/gpurun_out/trt_$N -name 't_kernel_trace.csv' | head -1)
  python3 $R/tools/trace_timeline.py $F > $R/gpurun_out/r5_tile_$N.txt 2>&1
  grep -- "--- " $R/gpurun_out/r5_tile_$N.txt
  rm -rf $R/gpurun_out/trt_$N
done; done
