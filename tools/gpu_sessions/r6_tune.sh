# round 6: autotuned launch choices against the cost model's : `r6_tune.sh`
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_latency.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do for T in 1 0; do for cfg in "f16x3 1 384" "f16x3 1 640" "fp32 1 384" "f16x3 4 384" "fp32 1 640" "f16x3 2 384"; do set -- $cfg; echo "WTK_SK_AUTOTUNE=$T $cfg: $(WTK_SK_AUTOTUNE=$T python3 tools/gpu_sessions/time_mode.py --dtype $1 --steps 300 --batch $2 --size $3 --plan latency 2>&1 | grep 'ms per step')"; done; done; done
