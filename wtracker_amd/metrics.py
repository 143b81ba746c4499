"""Track-comparison statistics used by the parity tests, tools/fp16_accuracy.py and bench.py's `parity` object:
survivor-index match rate and IoU distribution of one detector run against another on the same frames
("fp16 mode reports IoU distribution and index-match rate").  Pure numpy on host arrays;
the per-frame error the reference's own evaluation uses is `ErrorCalculator.calculate_bbox_error`
(wtracker/eval/error_calculator.py:163-195) — IoU is the detector-level counterpart BASELINE.json names."""
from __future__ import annotations

import numpy as np


def iou_xywh(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Row-wise IoU of two [N,4] (x, y, w, h) box arrays (float64)."""
    a = np.asarray(a, dtype=np.float64).reshape(-1, 4)
    b = np.asarray(b, dtype=np.float64).reshape(-1, 4)
    ax2, ay2, bx2, by2 = a[:, 0] + a[:, 2], a[:, 1] + a[:, 3], b[:, 0] + b[:, 2], b[:, 1] + b[:, 3]
    iw = np.clip(np.minimum(ax2, bx2) - np.maximum(a[:, 0], b[:, 0]), 0, None)
    ih = np.clip(np.minimum(ay2, by2) - np.maximum(a[:, 1], b[:, 1]), 0, None)
    inter = iw * ih
    u = a[:, 2] * a[:, 3] + b[:, 2] * b[:, 3] - inter
    return np.where(u > 0, inter / np.maximum(u, 1e-300), 1.0)


def _dist(v: np.ndarray):
    if len(v) == 0:
        return None
    return {"min": float(v.min()), "p01": float(np.percentile(v, 1)), "p05": float(np.percentile(v, 5)),
            "p50": float(np.percentile(v, 50)), "mean": float(v.mean())}


def accuracy_report(xywh, anchor, xywh_ref, anchor_ref, conf=None, conf_ref=None) -> dict:
    """(xywh [N,4], anchor [N]; anchor < 0 = NaN row) of a run against the same of a checker run."""
    anchor, anchor_ref = np.asarray(anchor), np.asarray(anchor_ref)
    xywh, xywh_ref = np.asarray(xywh, dtype=np.float64), np.asarray(xywh_ref, dtype=np.float64)
    det_r, det = anchor_ref >= 0, anchor >= 0
    both = det_r & det
    match = anchor == anchor_ref
    rep = {"frames": int(len(anchor_ref)), "checker_detections": int(det_r.sum()), "detections": int(det.sum()),
           "index_match": int(match.sum()), "index_match_rate": float(match.mean()) if len(match) else None,
           "nan_row_agreement": float((det_r == det).mean()) if len(match) else None,
           "iou_all_detected": _dist(iou_xywh(xywh[both], xywh_ref[both])),
           "iou_matched": _dist(iou_xywh(xywh[both & match], xywh_ref[both & match]))}
    if conf is not None and conf_ref is not None and both.any():
        rep["conf_abs_err_max"] = float(np.abs(np.asarray(conf)[both] - np.asarray(conf_ref)[both]).max())
    return rep
