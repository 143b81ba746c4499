# loader / consumer window kernel (WTK_HALO_LC=1) against the persistent kernel: bit-identity test, then per-op times of the layers it serves and frames/s
R=$GRAFT_REPO_ROOT
cd $R && timeout -k 10 300 python -m pytest tests/test_gpu_yolo.py -m gpu -q -x -k "loader_consumer" > gpurun_out/lc_tests.log 2>&1 || { tail -25 gpurun_out/lc_tests.log; exit 1; }
tail -2 gpurun_out/lc_tests.log
cd /tmp && export TMPDIR=/tmp
for lc in 0 1; do
  WTK_NO_SIDE_STREAM=1 WTK_HALO_LC=$lc timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/lc_tmp -o t -- python3 $R/tools/gpu_sessions/time_mode.py --dtype fp16 --steps 6 --batch 64 > $R/gpurun_out/lc_tmp.log 2>&1 || echo "trace failed"
  F=$(find $R/gpurun_out/lc_tmp -name 't_kernel_trace.csv' | head -1)
  python3 $R/tools/layer_profile.py $F --dtype fp16 --skip 3 --batch 64 > $R/gpurun_out/lc_layers_$lc.txt 2>&1
  echo "LC=$lc: $(grep -E '^model.6.m.0.cv1|^model.6.m.1.cv2|^model.12.m.0.cv2|^model.18.m.0.cv1|^model.8.m.0.cv1|^detect.0.cls|^TOTAL' $R/gpurun_out/lc_layers_$lc.txt | awk '{printf "%s %s us | ", $1, $3}')"
  rm -rf $R/gpurun_out/lc_tmp
done
cd $R
for round in 1 2; do for lc in 0 1; do
  WTK_HALO_LC=$lc python3 bench.py --dtype fp16 --no-fp32 --cpu-frames 0 --repeats 6 > gpurun_out/lc_b.json 2>/dev/null
  python3 -c "
import json
j=json.loads([l for l in open('gpurun_out/lc_b.json') if l.startswith('{')][-1]); r=j['roofline']
print('LC=$lc frames/s', round(j['value']), 'window family us/launch', round(r['avg_launch_ms']*1000,2), 'frac', round(r['frac'],4))"
done; done
