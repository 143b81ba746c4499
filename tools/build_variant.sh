#!/bin/bash
# Build the library under another name with extra compiler flags (A/B timing with tools/gpu_sessions/ab_lib.sh):
#   tools/build_variant.sh NAME "-DWTK_SILU_SCALAR_MASK=1"   ->  wtracker_amd/libwtk_hip_NAME.so   (git-ignored like every .so)
set -e
cd "$(dirname "$0")/.."
NAME=$1; FLAGS=$2
D=/tmp/wtk_variant_$NAME
rm -rf $D && mkdir -p $D/wtracker_amd $D/include && cp -r wtracker_amd/csrc $D/wtracker_amd/csrc && cp include/wtk_hip.h $D/include/
cd $D/wtracker_amd/csrc && rm -f *.o
objs=""
for f in wtk_api conv_igemm conv1x1_wide conv3x3_halo conv3x3_c32 front_fused front_fused_split c2f_fused stem_pool head mlp track_ops comm; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -Wall -Wno-unused-function $FLAGS -c $f.hip -o $f.o &
  objs="$objs $f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OLDPWD/wtracker_amd/libwtk_hip_$NAME.so $objs -ldl
echo built wtracker_amd/libwtk_hip_$NAME.so
