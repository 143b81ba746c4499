# single-stream kernel trace of the f16x3 forward (B=64, 640x640) -> per-op table gpurun_out/x3_layers_<TAG>.txt: `x3_trace.sh TAG [X3_BATCH=16] [ENV=VAL ...]`
R=$GRAFT_REPO_ROOT
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp WTK_NO_SIDE_STREAM=1
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trx_$TAG -o t -- python3 $R/tools/gpu_sessions/time_mode.py --dtype f16x3 --steps 6 --batch ${X3_BATCH:-64} > $R/gpurun_out/trx_$TAG.log 2>&1 || echo "trace failed"
F=$(find $R/gpurun_out/trx_$TAG -name 't_kernel_trace.csv' | head -1)
python3 $R/tools/layer_profile.py $F --dtype f16x3 --skip 3 --batch ${X3_BATCH:-64} > $R/gpurun_out/x3_layers_$TAG.txt 2>&1 || echo "layer table failed"
tail -3 $R/gpurun_out/x3_layers_$TAG.txt
