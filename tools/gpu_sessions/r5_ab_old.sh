R=$GRAFT_REPO_ROOT
cd $R
echo "== new (eager)"; python3 tools/gpu_sessions/r5_lat_time.py f16x3 latency 2>&1 | grep device
echo "== new (graph)"; WTK_GRAPH_VIEWS=1 python3 tools/gpu_sessions/r5_lat_time.py f16x3 latency 2>&1 | grep device
echo "== new (graph, all split)"; WTK_SK_FORM=0 WTK_GRAPH_VIEWS=1 python3 tools/gpu_sessions/r5_lat_time.py f16x3 latency 2>&1 | grep device
echo "== old b2877b1 (graph)"; WTK_HIP_LIB=$R/tools/_bin/libwtk_b2877b1.so python3 tools/gpu_sessions/r5_lat_time.py f16x3 latency 2>&1 | grep device
timeout -k 10 600 python3 -m pytest tests/test_gpu_latency.py -x -q -m gpu 2>&1 | tail -3
