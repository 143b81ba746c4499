#!/usr/bin/env python3
"""What decides how far fp16 moves a decision?  For every frame: the full-precision (f16x3) top-2 anchors a1, a2 and the fp16 noise of THEIR logit difference,
d = (L16(a1) - L16(a2)) - (Lx(a1) - Lx(a2)), split by the relation of the two anchors (same pyramid level and neighbouring cells / same level elsewhere /
different levels), plus the single-anchor noise L16(a1) - Lx(a1).  If neighbouring anchors' errors are strongly correlated, a margin threshold that knows the
relation could be much tighter for them than the global 6 sigma.
  python tools/pair_noise_study.py --frames 1024 [--weight-seed 0]
"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--seed", type=int, default=20000)
    ap.add_argument("--weight-seed", type=int, default=0)
    ap.add_argument("--size", type=int, default=640)
    args = ap.parse_args()
    from wtracker_amd import frames as fr, hip, yolo_spec as ys
    w = ys.synthetic_weights("s", 1, seed=args.weight_seed)
    depth, width, maxch = ys.SCALES["s"]
    B = 64
    mk = lambda dt: hip.HipYolo(w, (args.size, args.size), B, dtype=dt, nc=1, width=width, depth=depth, max_channels=maxch)
    d16, dx3 = mk("fp16"), mk("f16x3")
    frames = fr.diverse_frames(args.frames, args.size, seed=args.seed)
    S = args.size
    lv = [(S // 8, S // 8), (S // 16, S // 16), (S // 32, S // 32)]
    offs = np.cumsum([0] + [h * wd for h, wd in lv])

    def where(a):
        l = int(np.searchsorted(offs, a, side="right") - 1)
        j = a - offs[l]
        return l, j // lv[l][1], j % lv[l][1]

    rows = []
    for i in range(0, args.frames // B * B, B):
        d16.predict_host(frames[i:i + B], conf=0.1)
        c16 = d16.debug_head(B)[1][:, :, 0]
        dx3.predict_host(frames[i:i + B], conf=0.1)
        cx3 = dx3.debug_head(B)[1][:, :, 0]
        for n in range(B):
            o = np.argsort(-cx3[n])[:2]
            a1, a2 = int(o[0]), int(o[1])
            (l1, y1, x1), (l2, y2, x2) = where(a1), where(a2)
            rel = "adjacent" if l1 == l2 and max(abs(y1 - y2), abs(x1 - x2)) <= 1 else ("same_level" if l1 == l2 else "cross_level")
            rows.append((rel, float((c16[n, a1] - c16[n, a2]) - (cx3[n, a1] - cx3[n, a2])), float(c16[n, a1] - cx3[n, a1]), float(cx3[n, a1] - cx3[n, a2]),
                         int(np.argmax(c16[n]) != a1)))
    out = {}
    for rel in ("adjacent", "same_level", "cross_level", "all"):
        r = [x for x in rows if rel == "all" or x[0] == rel]
        if not r:
            continue
        d = np.array([x[1] for x in r]); e = np.array([x[2] for x in r]); g = np.array([x[3] for x in r]); bad = np.array([x[4] for x in r])
        out[rel] = {"frames": len(r), "pair_noise_std": float(d.std()), "pair_noise_maxabs": float(np.abs(d).max()), "single_noise_std": float(e.std()),
                    "gap_p10": float(np.percentile(g, 10)), "gap_p50": float(np.percentile(g, 50)), "fp16_flips": int(bad.sum()),
                    "share_gap_below_6sigma_own": float((g < 6 * d.std()).mean()), "share_gap_below_0.073": float((g < 0.073).mean())}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
