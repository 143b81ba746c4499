"""Ad-hoc sweep: default configuration (all fused / rescheduled kernels) against the plain layer-by-layer configuration on odd
shapes and batch sizes; every head logit and result must be bit-identical (fp16)."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from wtracker_amd import hip, yolo_spec as ys

OFF = {"WTK_NO_FUSED_FRONT": "1", "WTK_NO_FUSED_C2F": "1", "WTK_NO_FUSED_TAIL": "1", "WTK_MATERIALIZE_UPSAMPLE": "1",
       "WTK_HALO_SLABS": "2", "WTK_HALO_PERSIST": "0", "WTK_HALO_SMALL_BLOCKS": "0", "WTK_NO_WIDE_1X1": "1", "WTK_NO_IGEMM_TAIL": "1"}
w = ys.synthetic_weights("s", 1, seed=5)
depth, width, maxch = ys.SCALES["s"]
bad = 0
for (H, W), B, C in itertools.product([(32, 32), (64, 32), (32, 96), (224, 416), (416, 224), (608, 608), (640, 384), (1280, 736)], (1, 4, 7), (1, 3)):
    if H * W * B > 1280 * 736 * 4 and B == 7:
        continue
    rng = np.random.default_rng(H + W + B + C)
    frames = rng.integers(0, 256, size=(B, H, W) if C == 1 else (B, H, W, 3), dtype=np.uint8)
    outs = []
    for cfg in (OFF, {}):
        for k in OFF:
            os.environ.pop(k, None)
        os.environ.update(cfg)
        det = hip.HipYolo(w, (H, W), B, dtype="fp16", nc=1, width=width, depth=depth, max_channels=maxch)
        res = det.predict_host(frames, conf=0.05)
        outs.append((res, det.debug_head(B)))
        det.close()
    (ra, (ba, ca)), (rb, (bb, cb)) = outs
    same = np.array_equal(ba, bb) and np.array_equal(ca, cb) and all(np.array_equal(x, y, equal_nan=True) for x, y in zip(ra, rb))
    bad += not same
    print(("ok  " if same else "DIFF"), (H, W), "B", B, "C", C, flush=True)
print("mismatching configurations:", bad)
