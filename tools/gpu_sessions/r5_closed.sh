R=$GRAFT_REPO_ROOT
cd $R
for P in auto throughput latency; do python3 tools/gpu_sessions/r5_closed_probe.py $P 2>&1 | grep plan | cut -c1-70; done
echo "== threshold 4096"; for P in auto throughput; do WTK_SK_MIXED_MAX_PX=4096 python3 tools/gpu_sessions/r5_closed_probe.py $P 2>&1 | grep plan | cut -c1-70; done
