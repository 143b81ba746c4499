# HBM traffic of the f16x3 forward (B=64, 640x640): two PMC passes (FETCH_SIZE, WRITE_SIZE; no trace domains combined with --pmc), single stream.
#   traffic_x3.sh TAG [ENV=VAL ...]   ->  gpurun_out/pf_x3_fetch_TAG/, gpurun_out/pf_x3_write_TAG/  (read with tools/traffic_from_pmc.py)
R=$GRAFT_REPO_ROOT
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp WTK_NO_SIDE_STREAM=1
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pf_x3_fetch_$TAG -o p -- python3 $R/tools/gpu_sessions/time_mode.py --dtype f16x3 --steps 2 > $R/gpurun_out/pf_x3_fetch_$TAG.log 2>&1 || echo "fetch pass failed"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pf_x3_write_$TAG -o p -- python3 $R/tools/gpu_sessions/time_mode.py --dtype f16x3 --steps 2 > $R/gpurun_out/pf_x3_write_$TAG.log 2>&1 || echo "write pass failed"
ls $R/gpurun_out/pf_x3_fetch_$TAG $R/gpurun_out/pf_x3_write_$TAG
