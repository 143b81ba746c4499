R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp WTK_NO_SIDE_STREAM=1
for kv in "$@"; do export "$kv"; done
N=b15_thr_${TAG:-x}
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trb_$N -o t -- python3 $R/tools/gpu_sessions/time_mode.py --dtype ${DT:-f16x3} --steps 8 --batch 15 --size 384 --plan throughput > $R/gpurun_out/trb_$N.log 2>&1 || echo "trace failed"
F=$(find $R/gpurun_out/trb_$N -name 't_kernel_trace.csv' | head -1)
python3 $R/tools/trace_timeline.py $F > $R/gpurun_out/r5_$N.txt 2>&1
grep "ms per step" $R/gpurun_out/trb_$N.log; tail -12 $R/gpurun_out/r5_$N.txt
rm -rf $R/gpurun_out/trb_$N
