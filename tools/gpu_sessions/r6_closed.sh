# round 6: the closed-loop leg alone under A/B environments : `r6_closed.sh "ENV=VAL ..." "ENV=VAL ..." ...`  (each argument = one run; "-" = default environment)
R=$GRAFT_REPO_ROOT
cd $R
for envs in "$@"; do
  [ "$envs" = "-" ] && envs=""
  echo "== closed loop [$envs]"
  env $envs python3 bench.py --legs-child --no-latency > gpurun_out/r6_closed_tmp.out 2> gpurun_out/r6_closed_tmp.err || { echo failed; tail -5 gpurun_out/r6_closed_tmp.err; }
  python3 - <<'P'
import json
lines=[l for l in open("gpurun_out/r6_closed_tmp.out") if l.startswith("{")]
d=json.loads(lines[-1])["closed_loop"]
for k,v in d.items():
    if isinstance(v,dict) and "frames_per_s" in v and not k.startswith("cpu"):
        print(f"  {k:28s} {v['frames_per_s']:9.0f} frames/s  cycle {v['ms_per_cycle']:.3f} ms  B15 call {v['ms_cycle_batch_call_B15']}  B1 call {v['ms_single_frame_call_B1']:.3f}")
P
done
