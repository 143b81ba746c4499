R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_latency.py -x -q -m gpu 2>&1 | tail -2
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_bench2.out 2> gpurun_out/r5_bench2.err; echo bench rc $?; tail -c 2600 gpurun_out/r5_bench2.out; cp bench_detail.json gpurun_out/r5_bench2_detail.json
python3 - <<'P'
import json
d=json.load(open("bench_detail.json"))
for r in d["latency"]["rows"]: print({k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items() if k in ("dtype","plan","size","batch","device_ms","host_ms","pcie_ms")})
for k,v in d["closed_loop"].items():
    if isinstance(v,dict) and "frames_per_s" in v: print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})
    elif k=="error": print("ERROR", v)
P
