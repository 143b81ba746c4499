cd /tmp && export TMPDIR=/tmp WTK_NO_SIDE_STREAM=1
R=$GRAFT_REPO_ROOT
for v in 2 3; do
  WTK_HALO_SLABS=$v timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_s$v -o bench -- python3 $R/bench.py --steps 8 --warmup 3 --cpu-frames 0 --lanes 1 --no-profile > $R/gpurun_out/prof_s$v.log 2>&1 || echo fail $v
done
WTK_HALO_SLABS=3 timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcd_a3 -o p -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-profile --lanes 1 > $R/gpurun_out/pmcd_a3.log 2>&1 || echo fail pmc
echo done
