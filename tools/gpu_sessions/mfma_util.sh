# MFMA utilisation of the default workload's kernels: one PMC pass (no trace domains combined with --pmc), single stream
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp WTK_NO_SIDE_STREAM=1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pf_mfma -o p -- python3 $R/bench.py --steps 2 --warmup 1 --repeats 1 --dtype fp16 --no-fp32 --cpu-frames 0 --no-profile --lanes 1 > $R/gpurun_out/pf_mfma.log 2>&1 || echo "mfma pass failed"
ls $R/gpurun_out/pf_mfma
