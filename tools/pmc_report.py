#!/usr/bin/env python3
"""Per-op table of rocprofv3 --pmc counters (one or more counter_collection CSVs of bench.py runs).
  python tools/pmc_report.py gpurun_out/pmcA/*/*_counter_collection.csv [more.csv ...] [--filter halo]
"""
import collections
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.layer_profile import adapt_plan, plan  # noqa: E402


def load(path):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if "wtk" not in r["Kernel_Name"] or "mlp" in r["Kernel_Name"]:
            continue
        k = int(r["Dispatch_Id"])
        e = d.setdefault(k, {"name": r["Kernel_Name"], "dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        e[r["Counter_Name"]] = float(r["Counter_Value"])
    ks = sorted(d)
    per = len(adapt_plan(plan(), [d[k]["name"] for k in ks]))
    return [d[k] for k in ks[-per:]]


def main():
    files = [a for a in sys.argv[1:] if not a.startswith("--")]
    flt = None
    if "--filter" in sys.argv:
        flt = sys.argv[sys.argv.index("--filter") + 1]
    runs = [load(f) for f in files]
    ops = adapt_plan(plan(), [e["name"] for e in runs[0]])
    names = []
    for run in runs:
        for k in run[1]:
            if k not in ("name", "dur") and k not in names:
                names.append(k)
    print("op".ljust(24) + " ".join(n.replace("SQ_", "")[:14].rjust(14) for n in names))
    for i, (op, *_rest) in enumerate(ops):
        if flt and flt not in runs[0][i]["name"] and flt not in op:
            continue
        vals = {}
        for run in runs:
            vals.update(run[i])
        print(op.ljust(24) + " ".join(f"{vals.get(n, float('nan')):14.4g}" for n in names))


if __name__ == "__main__":
    main()
