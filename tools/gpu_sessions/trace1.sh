# single-stream kernel trace of the default workload -> per-kernel-name stats (gpurun_out/<tag>_stats.txt): `trace1.sh TAG [ENV=VAL ...]`
R=$GRAFT_REPO_ROOT
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp WTK_NO_SIDE_STREAM=1
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_$TAG -o t -- python3 $R/bench.py --steps 6 --warmup 2 --repeats 1 --cpu-frames 0 --no-fp32 --lanes 1 --no-profile > $R/gpurun_out/tr_$TAG.log 2>&1 || echo "trace failed"
python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/tr_$TAG/**/t_kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    n = n.replace("(anonymous namespace)::", ""); n = n[:n.index("(")] if "(" in n else n
    d[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = open("$R/gpurun_out/${TAG}_stats.txt", "w")
tot = sum(sum(v) for v in d.values())
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    line = "%9.1f us avg  %5d calls  %6.2f %%  %s" % (sum(v) / len(v) / 1e3, len(v), 100.0 * sum(v) / tot, n[-110:])
    print(line); out.write(line + "\n")
PY
