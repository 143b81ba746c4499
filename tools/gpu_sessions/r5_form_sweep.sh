R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_latency.py -x -q -m gpu > gpurun_out/r5_form_tests.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/r5_form_tests.log
run() { echo "== $*"; env "$@" python3 tools/gpu_sessions/r5_lat_time.py f16x3 latency 2>&1 | grep device; }
run WTK_X=0
run WTK_SK_FORM=0
run WTK_SK_FORM=1
run WTK_SK_FORM=0 WTK_SK_TILE=3
run WTK_SK_FORM=1 WTK_SK_TILE=0
run WTK_SK_FORM=1 WTK_SK_TILE=1
WTK_SK_VERBOSE=1 WTK_GRAPH_MAX_BATCH=0 python3 tools/gpu_sessions/time_mode.py --dtype f16x3 --steps 1 --batch 15 --size 384 --plan latency 2>&1 | grep conv_sk | tail -57 > gpurun_out/r5_form_choices_b15.txt
WTK_SK_VERBOSE=1 WTK_GRAPH_MAX_BATCH=0 python3 tools/gpu_sessions/time_mode.py --dtype f16x3 --steps 1 --batch 1 --size 384 --plan latency 2>&1 | grep conv_sk | tail -57 > gpurun_out/r5_form_choices_b1.txt
