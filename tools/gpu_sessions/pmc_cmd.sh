cd /tmp && export TMPDIR=/tmp WTK_NO_SIDE_STREAM=1
R=$GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmcd_$name -o p -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-profile --lanes 1 > $R/gpurun_out/pmcd_$name.log 2>&1 || echo "pass $name failed"; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE && \
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE && \
run c SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU && \
run d TA_BUSY_avr TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
ls $R/gpurun_out/pmcd_a | head -3
