#!/usr/bin/env python3
"""HBM traffic of the conv kernel family from two rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE) of
`bench.py --lanes 1 --no-profile` with WTK_NO_SIDE_STREAM=1 (single stream, launch order = plan order).
Applies the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE reports
half the bytes of wide coalesced reads -> doubled; WRITE_SIZE is exact.  Both counters are in KiB.

  python tools/traffic_from_pmc.py fetch.csv write.csv --batch 64 > profiles/r01_conv_traffic.json
"""
import argparse
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.layer_profile import adapt_plan, plan  # noqa: E402


def last_forward(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and "wtk" in r["Kernel_Name"] and "mlp" not in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    per = len(adapt_plan(plan(), [r["Kernel_Name"] for r in rows]))
    return rows[-per:]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("--batch", type=int, default=64)
    a = ap.parse_args()
    f = last_forward(a.fetch_csv, "FETCH_SIZE")
    ops = adapt_plan(plan(), [r["Kernel_Name"] for r in f])
    w = last_forward(a.write_csv, "WRITE_SIZE")
    total = 0.0
    n = 0
    per_op = {}
    per_kernel = {}
    for (name, kind, *_), rf, rw in zip(ops, f, w):
        b = 2.0 * float(rf["Counter_Value"]) * 1024 + float(rw["Counter_Value"]) * 1024
        per_op[name] = b
        if kind in ("conv", "fused"):
            total += b
            n += 1
        kn = rf["Kernel_Name"].replace("conv3x3_ws64", "conv3x3_halo").replace("conv3x3_s2", "conv_igemm")  # as wtk_yolo_get_kernel_profile counts them: the weight-stationary 64-channel form with the window kernel, the strided window kernel with the family of the implicit GEMM it replaces
        key = next((k for k in ("conv3x3_halo", "conv_igemm", "front_fused", "c2f32_fused", "conv3x3_c32", "sppf_pool", "stem_mfma") if k in kn), "head")
        if "conv1x1_wide" in kn:
            key = "conv_igemm"
        key = {"front_fused": "front_fused_kernel+c2f32_fused_kernel", "c2f32_fused": "front_fused_kernel+c2f32_fused_kernel", "head": "head",
               "conv_igemm": "conv_igemm_kernel+conv1x1_wide_kernel"}.get(key, key + "_kernel")
        e = per_kernel.setdefault(key, {"launches": 0, "hbm_bytes_per_forward": 0.0})
        e["launches"] += 1
        e["hbm_bytes_per_forward"] += b
    for e in per_kernel.values():
        e["hbm_bytes_per_launch_avg"] = e["hbm_bytes_per_forward"] / e["launches"]
    from wtracker_amd import _build

    print(json.dumps({"src_sha": _build.source_sha(), "collected": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, two passes, single stream", "conv_launches": n, "hbm_bytes_per_forward": total, "hbm_bytes_per_launch_avg": total / n,
                      "batch": a.batch, "correction": "FETCH_SIZE x2 (gfx950 wide-load under-count), WRITE_SIZE exact, KiB units",
                      "per_kernel": per_kernel, "per_op_bytes": per_op}, indent=1))


if __name__ == "__main__":
    main()
