# round 6: latency-plan tests, then kernel traces of one forward as timelines -> gpurun_out/r6_lat_<TAG>_<dtype>_b<B>_<S>.txt : `r6_lat.sh TAG [notest] [ENV=VAL ...]`
R=$GRAFT_REPO_ROOT
TAG=$1; shift
if [ "$1" = "notest" ]; then shift; else
  cd $R && timeout -k 10 900 python3 -m pytest tests/test_gpu_latency.py -x -q -m gpu > gpurun_out/r6_lat_${TAG}_tests.log 2>&1; echo "tests rc $?"; tail -15 gpurun_out/r6_lat_${TAG}_tests.log
fi
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
for cfg in "f16x3 1 384" "f16x3 1 640" "fp32 1 384" "f16x3 4 384"; do
  set -- $cfg; DT=$1; B=$2; S=$3
  N=${TAG}_${DT}_b${B}_${S}
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trl_$N -o t -- python3 $R/tools/gpu_sessions/time_mode.py --dtype $DT --steps 8 --batch $B --size $S --plan latency > $R/gpurun_out/trl_$N.log 2>&1 || { echo "trace $N failed"; tail -5 $R/gpurun_out/trl_$N.log; }
  F=$(find $R/gpurun_out/trl_$N -name 't_kernel_trace.csv' | head -1)
  python3 $R/tools/trace_timeline.py $F > $R/gpurun_out/r6_lat_$N.txt 2>&1 || echo "timeline $N failed"
  grep "ms per step" $R/gpurun_out/trl_$N.log
  grep -- "--- " $R/gpurun_out/r6_lat_$N.txt
  rm -rf $R/gpurun_out/trl_$N
done
# untraced device time per call, grouped against one launch per conv (same box, same process order)
cd $R
for G in 1 graph; do for cfg in "f16x3 1 384" "f16x3 15 384"; do set -- $cfg; GE="WTK_SK_GROUP=$G"; [ "$G" = graph ] && GE="WTK_GRAPH=1"; echo "$GE $cfg: $(env $GE python3 tools/gpu_sessions/time_mode.py --dtype $1 --steps 200 --batch $2 --size $3 --plan latency 2>&1 | grep 'ms per step')"; done; done
# the 32 x 32 tile against the four tiles of round 5
for T in 4 3; do for cfg in "f16x3 1 384" "f16x3 1 640" "fp32 1 384" "f16x3 4 384" "fp32 1 640"; do set -- $cfg; echo "WTK_SK_MAX_TILE=$T $cfg: $(WTK_SK_MAX_TILE=$T python3 tools/gpu_sessions/time_mode.py --dtype $1 --steps 200 --batch $2 --size $3 --plan latency 2>&1 | grep 'ms per step')"; done; done
