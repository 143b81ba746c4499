#!/usr/bin/env python3
"""One-off calibration of the per-conv gains used by wtracker_amd.yolo_spec.synthetic_weights.

Random weights through ~25 SiLU convs with residuals and concats drift in scale; a drifting net
saturates every class score at 1.0 and makes the arg-max parity test a test of ties.  This script
runs the CPU restatement once over a few synthetic frames and records, per conv, the factor that
makes the pre-activation standard deviation 1 (LSUV-style, single sequential pass), and for the
Detect heads' last 1x1 the factor that gives box logits std 1.5 and class logits std 2.5.
The factors are written to wtracker_amd/data/synth_gain_<scale>.json (63 floats) so that weight
generation itself stays deterministic and torch-free.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import yolo_oracle as yo  # noqa: E402
from wtracker_amd import frames as fr  # noqa: E402
from wtracker_amd import yolo_spec as ys  # noqa: E402


def calibrate(scale: str, size: int = 320, n_frames: int = 4, seed: int = 0) -> dict:
    depth, width, maxch = ys.SCALES[scale]
    w = ys.synthetic_weights(scale, 1, seed=seed, gains={})
    m = yo.YoloOracle(w, ys.model_dims(width, depth, maxch, 1))
    f, _ = fr.synthetic_frames(n_frames, size, seed=3)
    gains = {}

    def conv(name, x, stride=1, act=True):
        wt, b = m.w[name]
        pre = F.conv2d(x, wt, None, stride=stride, padding=wt.shape[2] // 2)
        target = 1.0
        if name.startswith("model.22.") and name.endswith(".2"):
            target = 1.5 if ".cv2." in name else 2.5
        g = target / float(pre.std())
        gains[name] = g
        y = pre * g + b.view(1, -1, 1, 1)
        return F.silu(y) if act else y

    m.conv = conv
    with torch.no_grad():
        x, _ = yo.preprocess(list(f), size)
        m.forward(x)
    return gains


if __name__ == "__main__":
    # default: the seed-0 tables of scales n and s;  `--seeds 1 2 3`: per-seed tables for scale s (synth_gain_s_seed<k>.json) — the factors are
    # a property of the weight DRAW (measured round 3: with the seed-0 table a seed-1 net saturates every class score at 1.0)
    os.makedirs(os.path.join(ROOT, "wtracker_amd", "data"), exist_ok=True)
    if "--seeds" in sys.argv:
        for seed in (int(v) for v in sys.argv[sys.argv.index("--seeds") + 1 :]):
            g = calibrate("s", seed=seed)
            path = os.path.join(ROOT, "wtracker_amd", "data", f"synth_gain_s_seed{seed}.json")
            json.dump(g, open(path, "w"), indent=0)
            print("s seed", seed, "gains min %.3f max %.3f" % (min(g.values()), max(g.values())), "->", path)
        sys.exit(0)
    for scale in ("n", "s"):
        g = calibrate(scale)
        path = os.path.join(ROOT, "wtracker_amd", "data", f"synth_gain_{scale}.json")
        json.dump(g, open(path, "w"), indent=0)
        print(scale, "gains min %.3f max %.3f" % (min(g.values()), max(g.values())), "->", path)
