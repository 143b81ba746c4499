# TIMING experiment (garbage results in the variant): the fp16 128-cout window kernels with 8 x v_mfma_f32_32x32x16_f16 per k-half instead of 16 x 16x16x32
# (tools/build_variant.sh mfma32 "-DWTK_TIMING_MFMA32").  Prints frames/s and the window family's average launch time for both libraries, alternating.
R=$GRAFT_REPO_ROOT
for round in 1 2 3; do
for L in wtracker_amd/libwtk_hip.so wtracker_amd/libwtk_hip_stagger.so; do
  WTK_HIP_LIB=$R/$L python3 $R/bench.py --dtype fp16 --no-fp32 --cpu-frames 0 --repeats 6 > $R/gpurun_out/abm_tmp.json 2>/dev/null
  python3 - <<PY
import json
j=json.loads([l for l in open('$R/gpurun_out/abm_tmp.json') if l.startswith('{')][-1])
r=j['roofline']
print('$L'.split('/')[-1], 'frames/s', round(j['value']), '| window family: us/launch', round(r['avg_launch_ms']*1000,2), 'frac', round(r['frac'],4), '| all conv us/launch', round(r['conv_family']['avg_launch_ms']*1000,2))
PY
done
done
