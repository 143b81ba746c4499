# round 6: wall ms per call of the latency plan (eager back-to-back calls) + dispatch counts : `r6_quick.sh`
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "f16x3 1 384" "f16x3 1 640" "fp32 1 384" "f16x3 4 384" "fp32 1 640" "f16x3 2 384"; do
  set -- $cfg; DT=$1; B=$2; S=$3; N=q_${DT}_b${B}_${S}
  echo "$cfg: $(python3 $R/tools/gpu_sessions/time_mode.py --dtype $DT --steps 300 --batch $B --size $S --plan latency 2>&1 | grep 'ms per step')"
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trl_$N -o t -- python3 $R/tools/gpu_sessions/time_mode.py --dtype $DT --steps 8 --batch $B --size $S --plan latency > $R/gpurun_out/trl_$N.log 2>&1 || echo "trace failed"
  python3 $R/tools/trace_timeline.py $(find $R/gpurun_out/trl_$N -name 't_kernel_trace.csv' | head -1) 2>/dev/null | grep -- "--- \|sk_finish"
  rm -rf $R/gpurun_out/trl_$N
done
