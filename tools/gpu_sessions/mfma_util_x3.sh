# MFMA utilisation of the f16x3 forward (B=64, 640x640): one PMC pass (no trace domains combined with --pmc), single stream
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp WTK_NO_SIDE_STREAM=1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pf_mfma_x3 -o p -- python3 $R/tools/gpu_sessions/time_mode.py --dtype f16x3 --steps 2 > $R/gpurun_out/pf_mfma_x3.log 2>&1 || echo "mfma pass failed"
ls $R/gpurun_out/pf_mfma_x3
