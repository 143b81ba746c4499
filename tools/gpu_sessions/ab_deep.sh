# TIMING ablations of the persistent window kernel's tap (garbage results): per-op time of the layers it serves, per variant library
# (tools/build_variant.sh skipN "-DWTK_TIMING_SKIP=N": 1 no weight-slab requests, 2 no window-piece requests, 4 no tap barrier)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp WTK_NO_SIDE_STREAM=1
for L in libwtk_hip.so libwtk_hip_deep.so libwtk_hip.so libwtk_hip_deep.so; do
  WTK_HIP_LIB=$R/wtracker_amd/$L timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ts_tmp -o t -- python3 $R/tools/gpu_sessions/time_mode.py --dtype fp16 --steps 6 --batch 64 > $R/gpurun_out/ts_tmp.log 2>&1 || echo "trace failed for $L"
  F=$(find $R/gpurun_out/ts_tmp -name 't_kernel_trace.csv' | head -1)
  python3 $R/tools/layer_profile.py $F --dtype fp16 --skip 3 --batch 64 > $R/gpurun_out/ts_layers_$L.txt 2>&1
  echo "$L: $(grep -E '^model.6.m.0.cv1|^model.12.m.0.cv2|^detect.0.0|^detect.1.0|^TOTAL' $R/gpurun_out/ts_layers_$L.txt | awk '{printf "%s %s us | ", $1, $3}')"
  rm -rf $R/gpurun_out/ts_tmp
done
