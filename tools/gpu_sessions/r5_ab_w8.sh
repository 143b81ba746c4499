R=$GRAFT_REPO_ROOT
cd $R
for L in cur w8; do
echo "== $L tile2 all split"; WTK_HIP_LIB=$R/tools/_bin/libwtk_$L.so WTK_SK_TILE=2 WTK_SK_FORM=0 WTK_GRAPH_VIEWS=1 python3 tools/gpu_sessions/r5_lat_time.py f16x3 latency 2>&1 | grep device
echo "== $L tile2 unsplit"; WTK_HIP_LIB=$R/tools/_bin/libwtk_$L.so WTK_SK_TILE=2 WTK_SK_FORM=1 WTK_GRAPH_VIEWS=1 python3 tools/gpu_sessions/r5_lat_time.py f16x3 latency 2>&1 | grep device
done
