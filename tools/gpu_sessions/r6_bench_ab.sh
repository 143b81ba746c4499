# round 6: bench line with the small-batch legs in a child process (default) and in-process; grouped vs one launch per conv at B = 2, 4 on the latency plan
R=$GRAFT_REPO_ROOT
cd $R
for G in 1 0; do for cfg in "f16x3 2 384" "f16x3 4 384" "fp32 4 384" "f16x3 2 640"; do set -- $cfg; echo "WTK_SK_GROUP=$G $cfg: $(WTK_SK_GROUP=$G python3 tools/gpu_sessions/time_mode.py --dtype $1 --steps 200 --batch $2 --size $3 --plan latency 2>&1 | grep 'ms per step')"; done; done
for MODE in child inprocess; do
  FLAG=""; [ $MODE = inprocess ] && FLAG="--legs-inprocess"
  timeout -k 10 600 python3 bench.py $FLAG > gpurun_out/r6_bench_$MODE.out 2> gpurun_out/r6_bench_$MODE.err; echo "bench $MODE rc $?"
  cp bench_detail.json gpurun_out/r6_bench_${MODE}_detail.json
  python3 - <<'P'
import json
d=json.load(open("bench_detail.json"))
print({k:round(v,3) for k,v in d.items() if k.startswith("value_") or k.startswith("latency_") or k.startswith("closed_loop_f")})
for k,v in (d.get("closed_loop") or {}).items():
    if isinstance(v,dict) and "frames_per_s" in v and not k.startswith("cpu"): print(f"  {k:28s} {v['frames_per_s']:9.0f} frames/s  cycle {v['ms_per_cycle']:.3f} ms  B15 {v['ms_cycle_batch_call_B15']}  B1 {v['ms_single_frame_call_B1']:.3f}")
P
done
