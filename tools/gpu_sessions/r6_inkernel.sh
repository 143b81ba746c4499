R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2; do for KB in 4096 6144 8192; do for cfg in "f16x3 1 384" "f16x3 1 640" "fp32 1 384" "f16x3 4 384"; do set -- $cfg; echo "WTK_SK_INKERNEL_MAX_KB=$KB $cfg: $(WTK_SK_INKERNEL_MAX_KB=$KB python3 tools/gpu_sessions/time_mode.py --dtype $1 --steps 300 --batch $2 --size $3 --plan latency 2>&1 | grep 'ms per step')"; done; done; done
