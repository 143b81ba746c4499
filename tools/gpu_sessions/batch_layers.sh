# Per-op time of the fp16 forward at several per-pass batch sizes (single stream): do the high-resolution layers get cheaper PER FRAME when a
# layer pair's tensors fit the XCDs' L2 (32 MiB) / the Infinity Cache (256 MiB)?   -> gpurun_out/bl_layers_<B>.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp WTK_NO_SIDE_STREAM=1
for B in 8 16 32 64; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/bl_$B -o t -- python3 $R/tools/gpu_sessions/time_mode.py --dtype fp16 --steps 6 --batch $B > $R/gpurun_out/bl_$B.log 2>&1 || echo "trace $B failed"
  F=$(find $R/gpurun_out/bl_$B -name 't_kernel_trace.csv' | head -1)
  python3 $R/tools/layer_profile.py $F --dtype fp16 --skip 3 --batch $B > $R/gpurun_out/bl_layers_$B.txt 2>&1 || echo "layer table $B failed"
  tail -1 $R/gpurun_out/bl_layers_$B.txt
  rm -rf $R/gpurun_out/bl_$B
done
