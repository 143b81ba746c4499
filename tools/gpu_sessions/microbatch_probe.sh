# Does a smaller per-pass batch (activations of a layer pair staying in the 256 MiB Infinity Cache) beat B = 64 per pass?  fp16 mode, frames/s of the default bench.
for cfg in "64 2" "32 2" "32 4" "16 4" "16 6" "8 6" "64 2"; do
  set -- $cfg
  python bench.py --dtype fp16 --no-fp32 --cpu-frames 0 --no-profile --batch $1 --lanes $2 --pool 128 --steps $((1280 / $1)) --repeats 6 > gpurun_out/mb_$1_$2.json 2>/dev/null
  python - <<PY
import json
j=json.load(open('gpurun_out/mb_$1_$2.json'))
print('batch', $1, 'lanes', $2, 'frames/s', round(j['value']), 'ms per 64 frames', round(64000/j['value'],3))
PY
done
