R=$GRAFT_REPO_ROOT
cd $R
run() { echo "== $*"; env "$@" python3 tools/gpu_sessions/r5_lat_time.py f16x3 latency 2>&1 | grep device; }
run WTK_X=0
run GPU_MAX_HW_QUEUES=8
run WTK_NO_SIDE_STREAM=1
run WTK_NO_SIDE_STREAM=1 GPU_MAX_HW_QUEUES=8
run WTK_GRAPH_MAX_BATCH=0 GPU_MAX_HW_QUEUES=8
run WTK_GRAPH_MAX_BATCH=0 WTK_NO_SIDE_STREAM=1
