#!/usr/bin/env python3
"""fp16-mode accuracy against the fp32 CPU restatement (oracle/yolo_oracle.py): survivor-index match rate
and IoU distribution (BASELINE.md §4 parity gate "fp16 mode reports IoU distribution and index-match rate").

  python tools/fp16_accuracy.py --frames 256 --size 640 --out gpurun_out/fp16_acc.json

Frames come from many seeds (consecutive frames of one synthetic track differ by half a pixel of worm
motion only, so one seed would be 256 nearly identical images).  Test infrastructure: imports oracle/.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seed", type=int, default=1000)
    ap.add_argument("--wseed", type=int, default=0)
    ap.add_argument("--conf", type=float, default=0.1)
    ap.add_argument("--out", default="")
    ap.add_argument("--dtypes", default="fp16,fp32,f16x3")
    args = ap.parse_args()

    from oracle import yolo_oracle as yo
    from wtracker_amd import frames as fr
    from wtracker_amd import hip
    from wtracker_amd import yolo_spec as ys
    from wtracker_amd.metrics import accuracy_report

    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    w = ys.synthetic_weights("s", 1, seed=args.wseed)
    depth, width, maxch = ys.SCALES["s"]
    oracle = yo.YoloOracle(w, ys.model_dims(width, depth, maxch, 1))
    frames = fr.diverse_frames(args.frames, args.size, args.seed)
    t0 = time.perf_counter()
    xo, co, ao, margin = [], [], [], []
    with torch.no_grad():
        for i in range(0, len(frames), 16):
            x, hw = yo.preprocess(list(frames[i : i + 16]), args.size)
            box, cls = oracle.forward(x)
            a, b, c = yo.postprocess(box, cls, tuple(x.shape[2:]), hw, conf=args.conf)
            xo.append(np.asarray(a, dtype=np.float64)), co.append(b), ao.append(c)
            top2 = torch.topk(cls.max(2).values, 2, dim=1).values  # logit gap between the best and second-best anchor
            margin.append((top2[:, 0] - top2[:, 1]).numpy())
    xo, co, ao, margin = np.concatenate(xo), np.concatenate(co), np.concatenate(ao), np.concatenate(margin)
    t_or = time.perf_counter() - t0
    rep = {"size": args.size, "weights_seed": args.wseed, "conf": args.conf, "oracle_seconds": t_or,
           "oracle_top1_top2_logit_gap": {"min": float(margin.min()), "p05": float(np.percentile(margin, 5)), "p50": float(np.percentile(margin, 50))}}
    for dtype in args.dtypes.split(","):
        det = hip.HipYolo(w, (args.size, args.size), args.batch, dtype=dtype, nc=1, width=width, depth=depth, max_channels=maxch)
        xg, cg, ag = [], [], []
        for i in range(0, len(frames), args.batch):
            a, b, c = det.predict_host(frames[i : i + args.batch], conf=args.conf)
            xg.append(a), cg.append(b), ag.append(c)
        det.close()
        rep[dtype] = accuracy_report(np.concatenate(xg), np.concatenate(ag), xo, ao, np.concatenate(cg), co)
        bad = np.nonzero(np.concatenate(ag) != ao)[0]
        rep[dtype]["mismatch_gap"] = [float(margin[i]) for i in bad[:32]]
    s = json.dumps(rep, indent=1)
    print(s)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        open(args.out, "w").write(s)


if __name__ == "__main__":
    main()
