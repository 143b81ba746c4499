# K atoms planned per handle (WTK_SK_PLAN_ATOMS): cycle batch timing, per-layer trace, then the latency / small-handle tests
R=$GRAFT_REPO_ROOT
cd $R
T="python3 tools/gpu_sessions/time_mode.py --size 384 --steps 300"
for rep in 1 2; do
for V in 0 1; do
  echo "== WTK_SK_PLAN_ATOMS=$V"
  WTK_SK_PLAN_ATOMS=$V $T --dtype f16x3 --batch 15 2>&1 | grep "ms per step"
  WTK_SK_PLAN_ATOMS=$V $T --dtype fp32 --batch 15 --steps 100 2>&1 | grep "ms per step"
  WTK_SK_PLAN_ATOMS=$V $T --dtype f16x3 --batch 9 --steps 200 2>&1 | grep "ms per step"
done
done
WTK_SK_VERBOSE=1 python3 tools/gpu_sessions/time_mode.py --size 384 --steps 1 --dtype f16x3 --batch 15 2>&1 | grep "conv_sk:" | sort | uniq -c | sort -rn | head -30
bash tools/gpu_sessions/r5_b15_trace.sh TAG=atoms1
timeout -k 10 600 python3 -m pytest tests/test_gpu_latency.py -x -q -m gpu 2>&1 | tail -3
