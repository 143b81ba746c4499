# conv_sk tuning sweep (device us per call, f16x3 latency plan): `r5_sk_sweep.sh`
R=$GRAFT_REPO_ROOT
cd $R
run() { echo "== $*"; env "$@" python3 tools/gpu_sessions/r5_lat_time.py f16x3 latency 2>&1 | grep device; }
run WTK_X=0
run WTK_SK_FINISH=1
run WTK_SK_SINGLE_MAX=18
run WTK_SK_SINGLE_MAX=24
run WTK_SK_SINGLE_MAX=36
run WTK_SK_SINGLE_MAX=24 WTK_SK_SLICE_STEPS=12
run WTK_SK_SINGLE_MAX=36 WTK_SK_SLICE_STEPS=18
run WTK_SK_SINGLE_MAX=12 WTK_SK_SLICE_STEPS=12
run WTK_SK_SINGLE_MAX=24 WTK_SK_FINISH=1
run WTK_SK_SINGLE_MAX=24 WTK_SK_INKERNEL_MAX_KB=65536
