#!/usr/bin/env python3
"""Map a rocprofv3 --kernel-trace CSV of bench.py back to the detector's ops (launch order is the
plan order of wtk_yolo_create) and print per-op time, TFLOP/s and algorithmic HBM GB/s.

  python tools/layer_profile.py gpurun_out/prof/.../*_kernel_trace.csv --batch 64 --size 640 --dtype fp16
"""
import argparse
import csv
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wtracker_amd import yolo_spec as ys  # noqa: E402


def plan(scale="s", nc=1):
    """(op name, kind, out stride, cout_stored, cin, k, extra output copies) in launch order."""
    depth, width, maxch = ys.SCALES[scale]
    d = ys.model_dims(width, depth, maxch, nc)
    c, n = d["c"], d["n"]
    ops = [("model.0", "stem", 2, c[0], 3, 3, 0)]

    def conv(name, s, cout, cin, k, extra=0):
        ops.append((name, "conv", s, cout, cin, k, extra))

    def c2f(p, s, c1, c2, nn, extra=0):
        cc = c2 // 2
        conv(p + ".cv1", s, 2 * cc, c1, 1)
        for i in range(nn):
            conv(f"{p}.m.{i}.cv1", s, cc, cc, 3)
            conv(f"{p}.m.{i}.cv2", s, cc, cc, 3)
        conv(p + ".cv2", s, c2, (2 + nn) * cc, 1, extra)

    conv("model.1", 4, c[1], c[0], 3)
    c2f("model.2", 4, c[1], c[1], n[0])
    conv("model.3", 8, c[2], c[1], 3)
    c2f("model.4", 8, c[2], c[2], n[1])
    conv("model.5", 16, c[3], c[2], 3)
    c2f("model.6", 16, c[3], c[3], n[2])
    conv("model.7", 32, c[4], c[3], 3)
    c2f("model.8", 32, c[4], c[4], n[3])
    conv("model.9.cv1", 32, c[4] // 2, c[4], 1)
    ops.append(("model.9.pool", "pool", 32, 3 * c[4] // 2, c[4] // 2, 5, 0))
    conv("model.9.cv2", 32, c[4], 2 * c[4], 1, 4)
    c2f("model.12", 16, c[4] + c[3], c[3], n[3], 4)
    c2f("model.15", 8, c[3] + c[2], c[2], n[3])
    conv("model.16", 16, c[2], c[2], 3)
    c2f("model.18", 16, c[2] + c[3], c[3], n[3])
    conv("model.19", 32, c[3], c[3], 3)
    c2f("model.21", 32, c[3] + c[4], c[4], n[3])
    ch = (c[2], c[3], c[4])
    for i, s in enumerate((8, 16, 32)):
        conv(f"detect.{i}.0(box+cls)", s, d["hb"] + d["hc"], ch[i], 3)
        conv(f"detect.{i}.box.1", s, d["hb"], d["hb"], 3)
        conv(f"detect.{i}.cls.1", s, d["hc"], d["hc"], 3)
        conv(f"detect.{i}.box.2", s, 64, d["hb"], 1)
        conv(f"detect.{i}.cls.2", s, 32, d["hc"], 1)
    ops.append(("head_select", "head", 8, 0, 0, 0, 0))
    return ops


def adapt_plan(ops, kernel_names, size=640, batch=64, es=2):
    """Replace the layer-by-layer entries by the fused launches a trace contains (fp16, YOLOv8s):
    front = model.0 + model.1 + model.2.cv1; C2f tail = model.2.m.0.cv1/cv2 + model.2.cv2.  The fused entries carry
    (flop, algorithmic bytes) in their last field."""
    if any("front_fused" in k for k in kernel_names):
        px4 = (size // 4) ** 2 * batch
        fl = sum(2.0 * (size // st) ** 2 * batch * co * ci * k * k for (nm, kd, st, co, ci, k, ex) in ops[:3])
        by = size * size * batch + px4 * ops[2][3] * es
        ops = [("front(model.0+1+2.cv1)", "fused", 4, 0, 0, 0, (fl, by))] + ops[3:]
    if any("c2f32_fused" in k for k in kernel_names):
        i0 = next(i for i, o in enumerate(ops) if o[0] == "model.2.m.0.cv1")
        px4 = (size // 4) ** 2 * batch
        fl = sum(2.0 * px4 * co * ci * k * k for (nm, kd, st, co, ci, k, ex) in ops[i0 : i0 + 3])
        by = px4 * 64 * es + px4 * 64 * es
        ops = ops[:i0] + [("model.2 tail(m.0+cv2)", "fused", 4, 0, 0, 0, (fl, by))] + ops[i0 + 3 :]
    # model.4.cv1 in the epilogue of model.3 (TAIL instantiation of the implicit-GEMM kernel: fourth of the five bools of its template list)
    if any("conv_igemm_kernel" in k and "Lb0ELb0ELb0ELb1ELb0EEEvNS_8ConvArgsE" in k for k in kernel_names):
        i0 = next(i for i, o in enumerate(ops) if o[0] == "model.3")
        (n3, k3, st3, co3, ci3, kk3, ex3), (n4, k4, st4, co4, ci4, kk4, ex4) = ops[i0], ops[i0 + 1]
        px = (size // st3) ** 2 * batch
        fl = 2.0 * px * co3 * ci3 * kk3 * kk3 + 2.0 * px * co4 * ci4
        by = px * 4 * ci3 * es + px * co4 * es
        ops = ops[:i0] + [("model.3+4.cv1", "fused", st3, 0, 0, 0, (fl, by))] + ops[i0 + 2 :]
    # Detect towers: the last 1x1 runs in the epilogue of the 3x3 before it (TAIL instantiations of the window kernel:
    # 64-cout tile = box tower, 128-cout tile = class tower)
    tails = {"box": any("conv3x3_halo_kernel" in k and "Li64E" in k and ("Lb1ELb0EEEvNS_8HaloArgsE" in k or "Lb1ELb1EEEvNS_8HaloArgsE" in k) for k in kernel_names),  # (TAIL, SPLIT) = (1, 0) fp16 / (1, 1) f16x3
             "cls": any("conv3x3_halo_kernel" in k and "Li128E" in k and ("Lb1ELb0EEEvNS_8HaloArgsE" in k or "Lb1ELb1EEEvNS_8HaloArgsE" in k) for k in kernel_names)}
    for tower, on in tails.items():
        if not on:
            continue
        out = []
        for o in ops:
            if o[0].startswith("detect.") and o[0].endswith(f".{tower}.2"):
                continue
            if o[0].startswith("detect.") and o[0].endswith(f".{tower}.1"):
                nm, kd, st, co, ci, k, ex = o
                px = (size // st) ** 2 * batch
                tail_out = 64 if tower == "box" else 32
                fl = 2.0 * px * co * ci * k * k + 2.0 * px * tail_out * co
                by = px * ci * es + px * tail_out * es
                o = (nm + "+2", "fused", st, 0, 0, 0, (fl, by))
            out.append(o)
        ops = out
    return ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--skip", type=int, default=2, help="forward passes to skip (warm-up)")
    args = ap.parse_args()
    es = 2 if args.dtype == "fp16" else 4
    rows = list(csv.DictReader(open(args.trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    mine = [r for r in rows if "wtk" in r["Kernel_Name"] and "mlp_kernel" not in r["Kernel_Name"]]
    ops = adapt_plan(plan(), [r["Kernel_Name"] for r in mine], args.size, args.batch, es)
    per = len(ops)
    n_fw = len(mine) // per
    assert n_fw > args.skip, f"{len(mine)} wtk dispatches, {per} per forward"
    acc = [0.0] * per
    cnt = 0
    for f in range(args.skip, n_fw):
        chunk = mine[f * per : (f + 1) * per]
        assert ("stem" in chunk[0]["Kernel_Name"] or "front_fused" in chunk[0]["Kernel_Name"]) and "head" in chunk[-1]["Kernel_Name"], "dispatch order does not match the plan"
        for i, r in enumerate(chunk):
            acc[i] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        cnt += 1
    tot_us = tot_fl = 0.0
    print(f"{'op':28s} {'kernel':10s} {'us':>9s} {'GFLOP':>8s} {'TF/s':>8s} {'MB(alg)':>9s} {'GB/s':>8s}")
    for i, (name, kind, s, cout, cin, k, extra) in enumerate(ops):
        us = acc[i] / cnt
        px = (args.size // s) ** 2 * args.batch
        if kind == "fused":
            fl, by = extra
        elif kind == "head":
            fl, by = 0.0, 0.0
        elif kind == "pool":
            fl, by = 0.0, px * (cin + cout) * es
        elif kind == "stem":
            fl, by = 2.0 * px * cout * 27, args.size * args.size * args.batch + px * cout * es
        else:
            stride_in = 2 if name in ("model.1", "model.3", "model.5", "model.7", "model.16", "model.19") else 1
            fl = 2.0 * px * cout * cin * k * k
            by = px * stride_in * stride_in * cin * es + px * cout * es * (1 + extra) + cout * cin * k * k * es
        kname = mine[args.skip * per + i]["Kernel_Name"]
        if kind == "fused":
            short = "fused"
        elif "c32" in kname:
            short = "c32"
        elif "conv1x1_wide" in kname:
            short = "wide256"
        elif "ws64" in kname:
            short = "ws64"
        elif "conv3x3_s2" in kname:
            short = "s2win"
        elif "halo" in kname:
            m = re.search(r"Li(\d+)ELi(\d)ELi(\d)E", kname)
            short = f"halo{m.group(1)}/{m.group(2)}" if m else "halo"
        else:
            short = "64x128" if "Li64ELi128E" in kname else "128x64" if "Li128ELi64E" in kname else "256x256" if "Li256ELi256" in kname else "128x128" if "Li128ELi128" in kname else ("256x64" if "Li256ELi64" in kname else ("256x32" if "Li256ELi32" in kname else kind))
        print(f"{name:28s} {short:10s} {us:9.1f} {fl / 1e9:8.2f} {fl / us / 1e6 if us else 0:8.1f} {by / 1e6:9.1f} {by / us / 1e3 if us else 0:8.0f}")
        tot_us += us
        tot_fl += fl
    print(f"{'TOTAL':28s} {'':10s} {tot_us:9.1f} {tot_fl / 1e9:8.2f} {tot_fl / tot_us / 1e6:8.1f}")


if __name__ == "__main__":
    main()
