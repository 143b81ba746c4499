# timing-only ablation of conv3x3_ws64_kernel (build with WTK_EXTRA_HIPCC_FLAGS=-DWTK_WS64_ABLATE): `ws64_ablate.sh "ABL:FLAGS ..."`
for pair in ${1:-0:3 0:11 0:19 0:27 1:3 7:3}; do
  abl=${pair%%:*}; fl=${pair##*:}
  bash tools/gpu_sessions/trace1.sh abl${abl}_${fl} WTK_WS64_ABLATE=$abl WTK_WS64_FLAGS=$fl > /dev/null 2>&1
  grep ws64 gpurun_out/abl${abl}_${fl}_stats.txt | sed "s/^/ABLATE=$abl FLAGS=$fl /"
done
