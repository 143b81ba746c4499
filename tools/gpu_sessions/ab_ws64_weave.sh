set -e
python -m pytest tests/test_gpu_yolo.py -m gpu -q -x -k "weight_stationary or fused_kernels_equal or full_size" > gpurun_out/r3_ws64_tests.log 2>&1 || { tail -30 gpurun_out/r3_ws64_tests.log; exit 1; }
tail -2 gpurun_out/r3_ws64_tests.log
for round in 1 2; do
for w in 0 1 2; do
  WTK_WS64_WEAVE=$w python bench.py --dtype fp16 --no-fp32 --cpu-frames 0 --repeats 6 > gpurun_out/r3_ws64_w${w}_${round}.json 2>/dev/null
  python - <<PY
import json
j=json.load(open('gpurun_out/r3_ws64_w${w}_${round}.json'))
k=[x for x in j['roofline']['kernels'] if 'ws64' in x['kernel']]
print('weave', $w, 'round', $round, 'value', round(j['value']), 'ws64 avg us', round(k[0]['avg_launch_ms']*1000,2) if k else None, 'TF', round(k[0]['achieved']) if k else None, 'dominant frac', round(j['roofline']['frac'],4))
PY
done
done
