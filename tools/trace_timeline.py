#!/usr/bin/env python3
"""One forward pass of a rocprofv3 --kernel-trace CSV as a timeline: per dispatch its start (us from the pass's first kernel), duration, the gap
since the previous dispatch ended, grid size in workgroups and a short kernel name; then totals per kernel.  The LAST complete pass of the trace
(front / stem kernel .. head kernel) is shown; `--pass -2` etc. pick earlier ones.

  python tools/trace_timeline.py gpurun_out/.../t_kernel_trace.csv [--pass -1] [--summary]
"""
import argparse
import csv
import re
from collections import OrderedDict


def short(name: str) -> str:
    m = re.search(r"(conv_sk_kernel|sk_finish_kernel|conv_igemm_kernel|conv3x3_halo_p?kernel|conv3x3_s2_kernel|conv3x3_ws64_kernel|conv3x3_c32_split_kernel|conv3x3_c32_kernel|conv1x1_wide_kernel|"
                  r"front_fused_split_kernel|front_fused_kernel|c2f32_fused_kernel|stem_mfma_kernel|sppf_pool_kernel|head_select_kernel|head_nms_kernel|view_letterbox_kernel|letterbox_kernel|mlp_kernel|[a-z0-9_]+_kernel)", name)
    base = m.group(1) if m else name[:40]
    t = re.search(r"ILb([01])ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", name)
    if base == "conv_sk_kernel" and t:
        base += f"<{'x3' if t.group(1) == '1' else 'f32'},{t.group(2)}x{t.group(3)},ns{t.group(6)}>"
    return base


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--pass", dest="which", type=int, default=-1)
    ap.add_argument("--summary", action="store_true", help="totals per kernel only")
    ap.add_argument("--queues", action="store_true", help="append the hardware queue id of every dispatch (concurrent side-stream runs)")
    args = ap.parse_args()
    rows = list(csv.DictReader(open(args.trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = [r for r in rows if "wtk" in r["Kernel_Name"]]
    starts = [i for i, r in enumerate(rows) if "front_fused" in r["Kernel_Name"] or "stem_mfma" in r["Kernel_Name"] or "letterbox" in r["Kernel_Name"]]
    # a pass = from a first kernel up to and including the next head kernel
    passes = []
    for i in starts:
        j = next((k for k in range(i, len(rows)) if "head_" in rows[k]["Kernel_Name"]), None)
        if j is not None and (not passes or i > passes[-1][1]):
            passes.append((i, j))
    i, j = passes[args.which]
    chunk = rows[i : j + 1]
    t0 = int(chunk[0]["Start_Timestamp"])
    prev_end = t0
    tot = OrderedDict()
    busy = 0.0
    for r in chunk:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        wg = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", "256")) or 256)
        gx = int(r.get("Grid_Size", r.get("Grid_Size_X", "0")) or 0)
        nm = short(r["Kernel_Name"])
        if not args.summary:
            q = f"  q{r['Queue_Id']}" if args.queues and "Queue_Id" in r else ""
            print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.2f}  gap {(s - prev_end) / 1e3:6.2f}  blocks {gx // max(wg, 1):6d}  {nm}{q}")
        d = tot.setdefault(nm, [0, 0.0])
        d[0] += 1
        d[1] += (e - s) / 1e3
        busy += (e - s) / 1e3
        prev_end = max(prev_end, e)
    span = (prev_end - t0) / 1e3
    print(f"--- {len(chunk)} dispatches, span {span:.1f} us, sum of durations {busy:.1f} us, {len(passes)} passes in the trace")
    for nm, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f"{us:9.1f} us  {n:4d} x  {nm}")


if __name__ == "__main__":
    main()
