# latency plan: tests, then single-stream kernel traces of one forward as timelines -> gpurun_out/r5_lat_<TAG>_<dtype>_b<B>_<S>.txt : `r5_lat.sh TAG [notest] [ENV=VAL ...]`
R=$GRAFT_REPO_ROOT
TAG=$1; shift
if [ "$1" = "notest" ]; then shift; else
  cd $R && timeout -k 10 600 python3 -m pytest tests/test_gpu_latency.py -x -q -m gpu > gpurun_out/r5_lat_${TAG}_tests.log 2>&1; echo "tests rc $?"; tail -15 gpurun_out/r5_lat_${TAG}_tests.log
fi
cd /tmp && export TMPDIR=/tmp WTK_NO_SIDE_STREAM=1
for kv in "$@"; do export "$kv"; done
for cfg in "f16x3 1 384" "f16x3 15 384" "f16x3 1 640" "fp32 1 384"; do
  set -- $cfg; DT=$1; B=$2; S=$3
  N=${TAG}_${DT}_b${B}_${S}
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trl_$N -o t -- python3 $R/tools/gpu_sessions/time_mode.py --dtype $DT --steps 8 --batch $B --size $S --plan latency > $R/gpurun_out/trl_$N.log 2>&1 || { echo "trace $N failed"; tail -5 $R/gpurun_out/trl_$N.log; }
  F=$(find $R/gpurun_out/trl_$N -name 't_kernel_trace.csv' | head -1)
  python3 $R/tools/trace_timeline.py $F > $R/gpurun_out/r5_lat_$N.txt 2>&1 || echo "timeline $N failed"
  grep "ms per step" $R/gpurun_out/trl_$N.log
  grep -- "--- " $R/gpurun_out/r5_lat_$N.txt
  rm -rf $R/gpurun_out/trl_$N
done
