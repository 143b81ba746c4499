# Profile artefacts of round 4 (tools/refresh_profiles_r04.sh copies them under profiles/): the default bench workload (batch 64, 640x640) in the HEADLINE
# mode (f16x3) and in the fp16 mode.  Every pass is the bench command itself (`python3 bench.py ...` right after `--`); PMC passes are separate runs with
# no trace domain next to --pmc.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-fp32 --cpu-frames 0 --no-check"
for M in f16x3 fp16; do
  # 1. kernel stats of the DEFAULT command's timed workload in this mode (2 lanes, side streams on)
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4pf_stats_$M -o bench -- $B --dtype $M --steps 10 --warmup 4 --repeats 2 > $R/gpurun_out/r4pf_stats_$M.log 2>&1 || echo "stats $M failed"
  # 2. single-stream trace for the per-layer table and the per-kernel averages the roofline object is checked against
  WTK_NO_SIDE_STREAM=1 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r4pf_layers_$M -o bench -- $B --dtype $M --steps 10 --warmup 4 --repeats 1 --lanes 1 --no-profile > $R/gpurun_out/r4pf_layers_$M.log 2>&1 || echo "layers $M failed"
  # 3. HBM traffic and MFMA-busy: separate PMC passes
  WTK_NO_SIDE_STREAM=1 timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r4pf_fetch_$M -o p -- $B --dtype $M --steps 2 --warmup 1 --repeats 1 --lanes 1 --no-profile > $R/gpurun_out/r4pf_fetch_$M.log 2>&1 || echo "fetch $M failed"
  WTK_NO_SIDE_STREAM=1 timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r4pf_write_$M -o p -- $B --dtype $M --steps 2 --warmup 1 --repeats 1 --lanes 1 --no-profile > $R/gpurun_out/r4pf_write_$M.log 2>&1 || echo "write $M failed"
  WTK_NO_SIDE_STREAM=1 timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r4pf_mfma_$M -o p -- $B --dtype $M --steps 2 --warmup 1 --repeats 1 --lanes 1 --no-profile > $R/gpurun_out/r4pf_mfma_$M.log 2>&1 || echo "mfma $M failed"
  echo "$M passes done"
done
ls $R/gpurun_out | grep r4pf_ | head -30
