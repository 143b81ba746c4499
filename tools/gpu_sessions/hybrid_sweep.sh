# hybrid mode (calibrated margin): lanes x defer sweep of the default bench, frames/s
for cfg in "2 5" "2 3" "2 8" "3 5" "3 8" "2 5"; do
  set -- $cfg
  python bench.py --dtype hybrid --no-fp32 --cpu-frames 0 --no-profile --lanes $1 --defer $2 --repeats 6 > gpurun_out/hs_$1_$2.json 2>/dev/null
  python - <<PY
import json
j=json.loads([l for l in open('gpurun_out/hs_$1_$2.json') if l.startswith('{')][-1])
print('lanes', $1, 'defer', $2, 'frames/s', round(j['value']), 'margin', round(j['hybrid_margin'],4), 'queue', j['hybrid_queue'], 'share', round(j['hybrid_second_look_share'],3), 'overflow', j['hybrid_overflow'])
PY
done
