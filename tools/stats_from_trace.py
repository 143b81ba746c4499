#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, min, max) of a rocprofv3 --kernel-trace CSV — the same columns as
rocprofv3 --stats, for traces taken without --stats (the single-stream trace behind the per-layer table).

  python tools/stats_from_trace.py gpurun_out/pf_layers/bench_kernel_trace.csv > profiles/r01_kernel_stats_single_stream_fp16_b64.csv
"""
import csv
import sys
from collections import defaultdict


def main():
    d = defaultdict(list)
    for r in csv.DictReader(open(sys.argv[1])):
        d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in d.values())
    w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k, len(v), sum(v), round(sum(v) / len(v), 1), round(100.0 * sum(v) / tot, 2), min(v), max(v)])


if __name__ == "__main__":
    main()
