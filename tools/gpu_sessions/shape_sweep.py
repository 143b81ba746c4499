"""Ad-hoc sweep: default configuration (all fused / rescheduled kernels) against the plain layer-by-layer configuration on odd
shapes and batch sizes; every head logit and result must be bit-identical (fp16).
`shape_sweep.py f16x3`: the split-fp16 handle — fused front off / on bit-identical; the thin-layer window kernel against the implicit GEMM
(another summation order) to 1e-4 of the logit scale with equal survivors."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from wtracker_amd import hip, yolo_spec as ys

OFF = {"WTK_NO_FUSED_FRONT": "1", "WTK_NO_FUSED_C2F": "1", "WTK_NO_FUSED_TAIL": "1", "WTK_MATERIALIZE_UPSAMPLE": "1",
       "WTK_HALO_SLABS": "2", "WTK_HALO_PERSIST": "0", "WTK_HALO_SMALL_BLOCKS": "0", "WTK_NO_WIDE_1X1": "1", "WTK_NO_IGEMM_TAIL": "1"}
DTYPE = sys.argv[1] if len(sys.argv) > 1 else "fp16"
if DTYPE == "f16x3":
    OFF = {"WTK_NO_FUSED_FRONT": "1"}
w = ys.synthetic_weights("s", 1, seed=5 if DTYPE == "fp16" else 0)
depth, width, maxch = ys.SCALES["s"]
bad = 0
for (H, W), B, C in itertools.product([(32, 32), (64, 32), (32, 96), (224, 416), (416, 224), (608, 608), (640, 384), (1280, 736)], (1, 4, 7), (1, 3)):
    if H * W * B > 1280 * 736 * 4 and B == 7:
        continue
    rng = np.random.default_rng(H + W + B + C)
    frames = rng.integers(0, 256, size=(B, H, W) if C == 1 else (B, H, W, 3), dtype=np.uint8)
    outs = []
    for cfg in (OFF, {}) + (({"WTK_NO_C32S": "1"},) if DTYPE == "f16x3" else ()):
        for k in list(OFF) + ["WTK_NO_C32S"]:
            os.environ.pop(k, None)
        os.environ.update(cfg)
        det = hip.HipYolo(w, (H, W), B, dtype=DTYPE, nc=1, width=width, depth=depth, max_channels=maxch)
        res = det.predict_host(frames, conf=0.05)
        outs.append((res, det.debug_head(B)))
        det.close()
    (ra, (ba, ca)), (rb, (bb, cb)) = outs[:2]
    same = np.array_equal(ba, bb) and np.array_equal(ca, cb) and all(np.array_equal(x, y, equal_nan=True) for x, y in zip(ra, rb))
    if DTYPE == "f16x3":
        rc, (bc, cc) = outs[2]
        sc = max(1.0, float(np.abs(cb).max()), float(np.abs(bb).max()))
        same = same and float(np.abs(cc - cb).max()) < 1e-4 * sc and float(np.abs(bc - bb).max()) < 1e-4 * sc and np.array_equal(rc[2], rb[2])
    bad += not same
    print(("ok  " if same else "DIFF"), (H, W), "B", B, "C", C, flush=True)
print("mismatching configurations:", bad)
