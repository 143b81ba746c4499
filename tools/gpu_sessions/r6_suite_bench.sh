# round 6: ONE ordinary run of the GPU suite, then the default bench line : `r6_suite_bench.sh TAG`
R=$GRAFT_REPO_ROOT
TAG=$1
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r6_suite_${TAG}.log 2>&1; echo "suite rc $?"; tail -6 gpurun_out/r6_suite_${TAG}.log
timeout -k 10 600 python3 bench.py > gpurun_out/r6_bench_${TAG}.out 2> gpurun_out/r6_bench_${TAG}.err; echo "bench rc $?"; tail -c 3000 gpurun_out/r6_bench_${TAG}.out; cp bench_detail.json gpurun_out/r6_bench_${TAG}_detail.json
python3 - <<'P'
import json
d=json.load(open("bench_detail.json"))
for r in (d.get("latency") or {}).get("rows", []): print({k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items() if k in ("dtype","plan","size","batch","device_ms","host_ms","pcie_ms")})
for k,v in (d.get("closed_loop") or {}).items():
    if isinstance(v,dict) and "frames_per_s" in v: print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})
    elif k=="error": print("ERROR", v)
P
