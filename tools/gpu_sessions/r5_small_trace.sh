# single-stream kernel traces of the small-batch forward (the reference's operating point: B = 1 / 15 at 384^2, BASELINE C2: B = 1 at 640^2)
# -> per-op tables gpurun_out/r5_small_<TAG>_<dtype>_b<B>_<S>.txt : `r5_small_trace.sh TAG [ENV=VAL ...]`
R=$GRAFT_REPO_ROOT
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp WTK_NO_SIDE_STREAM=1
for kv in "$@"; do export "$kv"; done
for cfg in "f16x3 1 384" "f16x3 15 384" "f16x3 1 640" "fp32 1 384"; do
  set -- $cfg; DT=$1; B=$2; S=$3
  N=${TAG}_${DT}_b${B}_${S}
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trs_$N -o t -- python3 $R/tools/gpu_sessions/time_mode.py --dtype $DT --steps 8 --batch $B --size $S > $R/gpurun_out/trs_$N.log 2>&1 || { echo "trace $N failed"; tail -5 $R/gpurun_out/trs_$N.log; }
  F=$(find $R/gpurun_out/trs_$N -name 't_kernel_trace.csv' | head -1)
  python3 $R/tools/layer_profile.py $F --dtype $DT --skip 4 --batch $B --size $S > $R/gpurun_out/r5_small_$N.txt 2>&1 || echo "layer table $N failed"
  grep "ms per step" $R/gpurun_out/trs_$N.log
  tail -1 $R/gpurun_out/r5_small_$N.txt
  rm -rf $R/gpurun_out/trs_$N
done
