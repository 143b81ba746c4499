R=$GRAFT_REPO_ROOT
cd $R
for NW in 1 0; do echo "== igemm narrow $NW"; WTK_IGEMM_NARROW=$NW python3 tools/gpu_sessions/r5_lat_time.py f16x3 throughput 2>&1 | grep device; done
WTK_LATENCY_PLAN=0 WTK_NO_SK_MIXED=1 python3 -m pytest tests/test_gpu_f16x3.py tests/test_gpu_yolo.py tests/test_gpu_latency.py -x -q -m gpu 2>&1 | tail -2
