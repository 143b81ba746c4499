# the cycle batch of 15 frames as k concurrent sub-batches on k handles / streams (time per step = all lanes' frames)
R=$GRAFT_REPO_ROOT
cd $R
T="python3 tools/gpu_sessions/time_mode.py --dtype f16x3 --size 384 --steps 300"
for rep in 1 2; do
$T --batch 15 --lanes 1 2>&1 | grep "ms per step"
$T --batch 8 --lanes 2 2>&1 | grep "ms per step"
$T --batch 5 --lanes 3 2>&1 | grep "ms per step"
$T --batch 4 --lanes 4 --plan throughput 2>&1 | grep "ms per step"
$T --batch 4 --lanes 4 --plan latency 2>&1 | grep "ms per step"
$T --batch 8 --lanes 1 2>&1 | grep "ms per step"
done
$T --dtype fp32 --batch 15 --lanes 1 2>&1 | grep "ms per step"
$T --dtype fp32 --batch 8 --lanes 2 2>&1 | grep "ms per step"
$T --dtype fp32 --batch 5 --lanes 3 2>&1 | grep "ms per step"
