# round 6: what the driver runs at round end — the GPU suite once, smoke(), the default bench line : `r6_final.sh TAG`
R=$GRAFT_REPO_ROOT
TAG=$1
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r6_final_${TAG}_suite.log 2>&1; echo "suite rc $?"; tail -4 gpurun_out/r6_final_${TAG}_suite.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6_final_${TAG}_smoke.log 2>&1; echo "smoke rc $?"; tail -5 gpurun_out/r6_final_${TAG}_smoke.log
timeout -k 10 600 python3 bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err; echo "bench rc $?"; tail -c 2500 gpurun_out/bench_default.log; cp bench_detail.json gpurun_out/r6_final_${TAG}_bench_detail.json
