"""Ad-hoc check: fused front vs layer-by-layer output of model.2.cv1 (conv index 2), where they differ."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wtracker_amd import hip, yolo_spec as ys  # noqa: E402

H, W, C, B = 128, 128, 1, 2
w = ys.synthetic_weights("s", 1, seed=0)
depth, width, maxch = ys.SCALES["s"]
rng = np.random.default_rng(5)
frames = rng.integers(0, 256, size=(B, H, W) if C == 1 else (B, H, W, 3), dtype=np.uint8)
outs = {}
os.environ["WTK_FRONT_DEBUG"] = "1"
for off in ("1", "0"):
    os.environ["WTK_NO_FUSED_FRONT"] = off
    det = hip.HipYolo(w, (H, W), B, dtype="fp16", nc=1, width=width, depth=depth, max_channels=maxch)
    det.predict_host(frames, conf=0.05)
    outs[off] = det.debug_tensor(2, B)
    outs["t0" + off], outs["t1" + off] = det.debug_tensor(0, B), det.debug_tensor(1, B)
    del det
for nm in ("t0", "t1"):
    x, y = outs[nm + "1"], outs[nm + "0"]
    dd = x != y
    print(nm, x.shape, "mismatch", dd.sum(), "max abs", np.abs(x - y).max())
    for i in np.argwhere(dd)[:12]:
        print("   ", tuple(int(v) for v in i), x[tuple(i)], y[tuple(i)])
a, b = outs["1"], outs["0"]
d = a != b
print("shape", a.shape, "mismatch", d.sum(), "of", d.size, "max abs", np.abs(a - b).max())
print("by image", d.sum(axis=(1, 2, 3)))
print("by row", d.sum(axis=(0, 2, 3)))
print("by col", d.sum(axis=(0, 1, 3)))
print("by channel", d.sum(axis=(0, 1, 2)))
idx = np.argwhere(d)[:10]
for i in idx:
    print(tuple(i), a[tuple(i)], b[tuple(i)])
