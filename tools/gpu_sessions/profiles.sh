# Profile artefacts (tools/refresh_profiles.sh copies them under profiles/) of the default bench workload (fp16, batch 64, 640x640)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
# 1. kernel stats of the DEFAULT bench command (2 lanes, side stream on)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pf_stats -o bench -- python3 $R/bench.py --steps 10 --warmup 4 --repeats 2 --dtype fp16 --no-fp32 --cpu-frames 0 > $R/gpurun_out/pf_stats.log 2>&1 || echo "stats failed"
# 1b. kernel stats of the HEADLINE mode of the default command (hybrid: the fp16 pass + the deferred f16x3 second looks, 2 lanes; --hybrid-margin fixed at the value the
#     calibration finds on the seed-0 weights so that the profiled run needs no calibration pass)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pf_stats_hyb -o bench -- python3 $R/bench.py --steps 10 --warmup 4 --repeats 2 --dtype hybrid --hybrid-margin 0.0731 --hybrid-queue 195 --no-fp32 --cpu-frames 0 --no-profile > $R/gpurun_out/pf_stats_hyb.log 2>&1 || echo "hybrid stats failed"
# 2. single-stream trace for the per-layer table
WTK_NO_SIDE_STREAM=1 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pf_layers -o bench -- python3 $R/bench.py --steps 10 --warmup 4 --repeats 1 --dtype fp16 --no-fp32 --cpu-frames 0 --lanes 1 --no-profile > $R/gpurun_out/pf_layers.log 2>&1 || echo "layers failed"
# 3. HBM traffic: separate PMC passes (no trace domains combined with --pmc)
WTK_NO_SIDE_STREAM=1 timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pf_fetch -o p -- python3 $R/bench.py --steps 2 --warmup 1 --repeats 1 --dtype fp16 --no-fp32 --cpu-frames 0 --lanes 1 --no-profile > $R/gpurun_out/pf_fetch.log 2>&1 || echo "fetch failed"
WTK_NO_SIDE_STREAM=1 timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pf_write -o p -- python3 $R/bench.py --steps 2 --warmup 1 --repeats 1 --dtype fp16 --no-fp32 --cpu-frames 0 --lanes 1 --no-profile > $R/gpurun_out/pf_write.log 2>&1 || echo "write failed"
echo done
