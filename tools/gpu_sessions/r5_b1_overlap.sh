# B = 1 on the latency plan WITH the side streams (the Detect towers of P3 / P4 beside the PAN path): where do the towers land?
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
N=b1_overlap_${TAG:-x}
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tro_$N -o t -- python3 $R/tools/gpu_sessions/time_mode.py --dtype f16x3 --steps 20 --batch 1 --size 384 --plan latency > $R/gpurun_out/tro_$N.log 2>&1 || echo "trace failed"
F=$(find $R/gpurun_out/tro_$N -name 't_kernel_trace.csv' | head -1)
python3 $R/tools/trace_timeline.py $F --queues > $R/gpurun_out/r5_$N.txt 2>&1
grep "ms per step" $R/gpurun_out/tro_$N.log; tail -8 $R/gpurun_out/r5_$N.txt
rm -rf $R/gpurun_out/tro_$N
