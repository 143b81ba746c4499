#!/usr/bin/env python3
"""Which decision margins do the fp16 mode's survivor mismatches have?  For seeded frame sets: the fp16 margin (wtk_yolo_last_margins_host) of every
frame whose survivor differs from the fp32 restatement's, and the share of frames below candidate re-check thresholds.
Test infrastructure (imports oracle/).   python tools/margin_study.py --sets 1000:256,5000:512
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", default="1000:256,5000:512")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--conf", type=float, default=0.1)
    args = ap.parse_args()
    from oracle import yolo_oracle as yo
    from wtracker_amd import frames as fr
    from wtracker_amd import hip
    from wtracker_amd import yolo_spec as ys

    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    w = ys.synthetic_weights("s", 1, seed=0)
    depth, width, maxch = ys.SCALES["s"]
    oracle = yo.YoloOracle(w, ys.model_dims(width, depth, maxch, 1))
    det = hip.HipYolo(w, (args.size, args.size), 64, dtype="fp16", nc=1, width=width, depth=depth, max_channels=maxch)
    rep = {}
    for spec in args.sets.split(","):
        seed, n = (int(v) for v in spec.split(":"))
        frames = fr.diverse_frames(n, args.size, seed)
        ao = []
        with torch.no_grad():
            for i in range(0, n, 16):
                x, hw = yo.preprocess(list(frames[i : i + 16]), args.size)
                box, cls = oracle.forward(x)
                ao.append(yo.postprocess(box, cls, tuple(x.shape[2:]), hw, conf=args.conf)[2])
        ao = np.concatenate(ao)
        ag, mg = [], []
        for i in range(0, n, 64):
            ag.append(det.predict_host(frames[i : i + 64], conf=args.conf)[2])
            mg.append(det.last_margins(len(frames[i : i + 64])))
        ag, mg = np.concatenate(ag), np.concatenate(mg)
        bad = ag != ao
        rep[spec] = {"frames": n, "mismatches": int(bad.sum()), "mismatch_margins": sorted(float(v) for v in mg[bad]),
                     "share_below": {str(t): float((mg < t).mean()) for t in (0.02, 0.03, 0.04, 0.05, 0.06, 0.08)},
                     "max_weak_per_64": {str(t): int(max((mg[i : i + 64] < t).sum() for i in range(0, n, 64))) for t in (0.04, 0.05, 0.06, 0.08)}}
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
