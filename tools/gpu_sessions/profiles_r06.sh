# Profile artefacts of round 6 (tools/refresh_profiles_r06.sh copies them under profiles/): the default bench workload (batch 64, 640x640) in the HEADLINE
# mode (f16x3) and in the fp16 mode.  Every pass is the bench command itself (`python3 bench.py ...` right after `--`); PMC passes are separate runs with
# no trace domain next to --pmc.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-fp32 --cpu-frames 0 --no-check"
for M in f16x3 fp16; do
  # 1. kernel stats of the DEFAULT command's timed workload in this mode (2 lanes, side streams on)
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6pf_stats_$M -o bench -- $B --dtype $M --steps 10 --warmup 4 --repeats 2 > $R/gpurun_out/r6pf_stats_$M.log 2>&1 || echo "stats $M failed"
  # 2. single-stream trace for the per-layer table and the per-kernel averages the roofline object is checked against
  WTK_NO_SIDE_STREAM=1 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r6pf_layers_$M -o bench -- $B --dtype $M --steps 10 --warmup 4 --repeats 1 --lanes 1 --no-profile > $R/gpurun_out/r6pf_layers_$M.log 2>&1 || echo "layers $M failed"
  # 3. HBM traffic and MFMA-busy: separate PMC passes
  WTK_NO_SIDE_STREAM=1 timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r6pf_fetch_$M -o p -- $B --dtype $M --steps 2 --warmup 1 --repeats 1 --lanes 1 --no-profile > $R/gpurun_out/r6pf_fetch_$M.log 2>&1 || echo "fetch $M failed"
  WTK_NO_SIDE_STREAM=1 timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r6pf_write_$M -o p -- $B --dtype $M --steps 2 --warmup 1 --repeats 1 --lanes 1 --no-profile > $R/gpurun_out/r6pf_write_$M.log 2>&1 || echo "write $M failed"
  WTK_NO_SIDE_STREAM=1 timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r6pf_mfma_$M -o p -- $B --dtype $M --steps 2 --warmup 1 --repeats 1 --lanes 1 --no-profile > $R/gpurun_out/r6pf_mfma_$M.log 2>&1 || echo "mfma $M failed"
  echo "$M passes done"
done
ls $R/gpurun_out | grep r6pf_ | head -30
# 4. the latency plan: kernel stats + timeline of the single-frame call (B = 1 at 384^2, f16x3 and fp32) and of B = 1 at 640^2 (BASELINE config 2)
for cfg in "f16x3 1 384" "fp32 1 384" "f16x3 1 640" "fp32 1 640" "f16x3 4 384"; do
  set -- $cfg; DT=$1; BB=$2; S=$3
  N=${DT}_b${BB}_${S}
  WTK_NO_SIDE_STREAM=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6pf_lat_$N -o lat -- python3 $R/tools/gpu_sessions/time_mode.py --dtype $DT --steps 20 --batch $BB --size $S --plan latency > $R/gpurun_out/r6pf_lat_$N.log 2>&1 || echo "latency trace $N failed"
done
# 5. the cycle batch on the plan the controller gives it: a throughput-plan handle of 15 frames (small-handle rules: split-K on the 12x12 maps, 64-cout tiles, six-slab ring)
WTK_NO_SIDE_STREAM=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6pf_thr_f16x3_b15_384 -o lat -- python3 $R/tools/gpu_sessions/time_mode.py --dtype f16x3 --steps 20 --batch 15 --size 384 --plan throughput > $R/gpurun_out/r6pf_thr_f16x3_b15_384.log 2>&1 || echo "cycle batch trace failed"
ls $R/gpurun_out | grep r6pf_ | head -40
# 6. BASELINE config 5's per-GPU shape (1280 x 1280, fp16, batch 256): the bench line and its MFMA-busy PMC pass
cd $R && timeout -k 10 500 python3 bench.py --size 1280 --batch 256 --dtype fp16 --steps 4 --warmup 1 --repeats 3 --no-fp32 --cpu-frames 0 --no-closed-loop --no-latency > gpurun_out/r6pf_bench_1280_b256.out 2> gpurun_out/r6pf_bench_1280_b256.err; echo "1280 bench rc $?"; tail -c 600 gpurun_out/r6pf_bench_1280_b256.out
cd /tmp && WTK_NO_SIDE_STREAM=1 timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r6pf_mfma_1280 -o p -- python3 $R/bench.py --size 1280 --batch 256 --dtype fp16 --steps 1 --warmup 1 --repeats 1 --lanes 1 --no-profile --no-fp32 --cpu-frames 0 --no-check --no-closed-loop --no-latency > $R/gpurun_out/r6pf_mfma_1280.log 2>&1 || echo "mfma 1280 failed"
# 7. the artefacts are rebuilt ON the box (so that the bench line that follows quotes this session's PMC files: provenance = these sources), then the default
#    bench line, then the artefacts again (with the line) into gpurun_out/profiles_r06/ for the way back
cd $R && bash tools/refresh_profiles_r06.sh > gpurun_out/r6pf_refresh.log 2>&1; echo "refresh rc $?"
timeout -k 10 600 python3 bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err; echo "default bench rc $?"; tail -c 400 gpurun_out/bench_default.log
bash tools/refresh_profiles_r06.sh >> gpurun_out/r6pf_refresh.log 2>&1
mkdir -p gpurun_out/profiles_r06 && cp profiles/r06_* gpurun_out/profiles_r06/ && ls gpurun_out/profiles_r06 | wc -l
