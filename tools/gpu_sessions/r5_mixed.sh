R=$GRAFT_REPO_ROOT
cd $R
for PX in 0 4096 10000 40000 200000; do echo "== mixed max px $PX"; WTK_SK_MIXED_MAX_PX=$PX python3 tools/gpu_sessions/r5_lat_time.py f16x3,fp32 throughput 2>&1 | grep device; done
