R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_latency.py -x -q -m gpu > gpurun_out/r5_check_tests.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/r5_check_tests.log
python3 tools/gpu_sessions/r5_lat_time.py f16x3,fp32 latency 2>&1 | grep device
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_bench1.out 2> gpurun_out/r5_bench1.err; echo bench rc $?; tail -c 3000 gpurun_out/r5_bench1.out; cp bench_detail.json gpurun_out/r5_bench1_detail.json
