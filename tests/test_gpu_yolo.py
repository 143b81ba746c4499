"""GPU parity: YOLOv8 detector path (stem, implicit-GEMM convs, SPPF, head select) vs the CPU
restatement (oracle/yolo_oracle.py; parity unpinned — no ultralytics, no trained weights) on the
same seeded weights and frames.  Everything goes through the C ABI."""
import os

import numpy as np
import pytest
import torch

from oracle import yolo_oracle as yo
from wtracker_amd import frames as fr
from wtracker_amd import hip
from wtracker_amd import yolo_spec as ys

pytestmark = pytest.mark.gpu

# fp32 mode: exact-fp32 MFMA vs torch-CPU fp32 — only summation order and expf differ
F32_LOGIT_ATOL = 2e-3
F32_BOX_ATOL = 2e-2   # pixels
# fp16 mode: fp16 storage of every activation; logits are O(1..10)
F16_LOGIT_ATOL = 0.35
F16_IOU_MIN = 0.90


def _iou_xywh(a, b):
    ax2, ay2, bx2, by2 = a[0] + a[2], a[1] + a[3], b[0] + b[2], b[1] + b[3]
    iw = max(0.0, min(ax2, bx2) - max(a[0], b[0]))
    ih = max(0.0, min(ay2, by2) - max(a[1], b[1]))
    inter = iw * ih
    u = a[2] * a[3] + b[2] * b[3] - inter
    return inter / u if u > 0 else 1.0


def _models(scale, size, dtype, nc=1, seed=0, max_batch=8):
    w = ys.synthetic_weights(scale, nc, seed=seed)
    depth, width, maxch = ys.SCALES[scale]
    oracle = yo.YoloOracle(w, ys.model_dims(width, depth, maxch, nc))
    det = hip.HipYolo(w, (size, size), max_batch, dtype=dtype, nc=nc, width=width, depth=depth, max_channels=maxch)
    return oracle, det


def _oracle_heads(oracle, frames, size):
    with torch.no_grad():
        x, hw = yo.preprocess(list(frames), size)
        box, cls = oracle.forward(x)
    return box, cls, hw


@pytest.mark.parametrize("scale,size,B", [("n", 128, 3), ("n", 160, 2), ("s", 128, 2), ("s", 256, 2)])
def test_fp32_head_logits_and_boxes_match_oracle(hip_lib, scale, size, B):
    oracle, det = _models(scale, size, "fp32")
    frames, _ = fr.synthetic_frames(B, size, seed=11)
    box_o, cls_o, hw = _oracle_heads(oracle, frames, size)
    xywh, conf, anchor = det.predict_host(frames, conf=0.1)
    box_g, cls_g = det.debug_head(B)
    np.testing.assert_allclose(cls_g, cls_o.numpy(), rtol=1e-3, atol=F32_LOGIT_ATOL)
    np.testing.assert_allclose(box_g, box_o.numpy(), rtol=1e-3, atol=F32_LOGIT_ATOL)
    xywh_o, conf_o, anchor_o = yo.postprocess(box_o, cls_o, (size, size), hw, conf=0.1)
    np.testing.assert_array_equal(anchor, anchor_o)  # survivor indices
    np.testing.assert_allclose(xywh, xywh_o, rtol=0, atol=F32_BOX_ATOL)
    np.testing.assert_allclose(conf, conf_o, rtol=0, atol=1e-4)


@pytest.mark.parametrize("size,B", [(128, 2), (256, 3)])
def test_fp16_matches_oracle_within_stated_tolerance(hip_lib, size, B):
    oracle, det = _models("s", size, "fp16")
    frames, _ = fr.synthetic_frames(B, size, seed=12)
    box_o, cls_o, hw = _oracle_heads(oracle, frames, size)
    xywh, conf, anchor = det.predict_host(frames, conf=0.1)
    box_g, cls_g = det.debug_head(B)
    assert np.abs(cls_g - cls_o.numpy()).max() < F16_LOGIT_ATOL
    assert np.abs(box_g - box_o.numpy()).max() < F16_LOGIT_ATOL
    # selection computed by the GPU from ITS OWN logits must equal the oracle's selection logic applied
    # to the same logits (bit-exact index), and be close to the fp32 oracle's boxes
    xywh_s, conf_s, anchor_s = yo.postprocess(torch.from_numpy(box_g), torch.from_numpy(cls_g), (size, size), hw, conf=0.1)
    np.testing.assert_array_equal(anchor, anchor_s)
    np.testing.assert_allclose(xywh, xywh_s, rtol=0, atol=2e-2)
    xywh_o, _, anchor_o = yo.postprocess(box_o, cls_o, (size, size), hw, conf=0.1)
    for n in range(B):
        if anchor[n] == anchor_o[n] and anchor[n] >= 0:
            assert _iou_xywh(xywh[n], xywh_o[n]) > F16_IOU_MIN


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_decode_and_selection_bit_exact_on_given_logits(hip_lib, dtype):
    """Isolates box-decode / arg-max selection from conv rounding: identical logits in, survivor index
    must be identical, including ties (lowest anchor wins), the conf threshold, NaN rows, clipping."""
    size, B = 128, 6
    _, det = _models("n", size, dtype)
    A = det.anchors
    rng = np.random.default_rng(4)
    box = rng.normal(0, 2, size=(B, A, 64)).astype(np.float16).astype(np.float32)  # exactly representable in both modes
    cls = (rng.normal(-4, 2, size=(B, A, 1))).astype(np.float16).astype(np.float32)
    cls[1, :, 0] = -6.0           # nothing above conf -> NaN row
    cls[2, 37, 0] = cls[2, 200, 0] = cls[2, 299, 0] = 5.0   # three-way tie -> lowest index 37
    cls[3, A - 1, 0] = 7.0        # last anchor (stride 32 level, corner -> clipping)
    cls[4, 0, 0] = 7.0            # first anchor (corner -> clipping at 0)
    box[5] = 0.0                  # uniform DFL -> every side 7.5 bins
    cls[5, 100, 0] = 3.0
    xywh, conf, anchor = det.decode_host(box, cls, size, size, conf=0.25)
    xywh_o, conf_o, anchor_o = yo.postprocess(torch.from_numpy(box), torch.from_numpy(cls), (size, size), (size, size), conf=0.25)
    np.testing.assert_array_equal(anchor, anchor_o)
    assert anchor[1] == -1 and np.isnan(xywh[1]).all() and anchor[2] == 37 and anchor[3] == A - 1 and anchor[4] == 0
    ok = anchor >= 0
    np.testing.assert_allclose(xywh[ok], xywh_o[ok], rtol=0, atol=1e-3)
    np.testing.assert_allclose(conf, conf_o, rtol=0, atol=1e-6)
    # zero box head: w = h = 15 * stride centred on the anchor (SURVEY.md §8c self-check), here stride 8
    lw = size // 8
    ax, ay = (100 % lw + 0.5) * 8, (100 // lw + 0.5) * 8
    x1, y1, x2, y2 = max(ax - 60, 0), max(ay - 60, 0), min(ax + 60, size), min(ay + 60, size)
    np.testing.assert_allclose(xywh[5], [x1, y1, x2 - x1, y2 - y1], atol=1e-3)


def test_gray_and_bgr_inputs_agree_and_channel_order(hip_lib):
    size, B = 128, 2
    oracle, det = _models("n", size, "fp32")
    gray, _ = fr.synthetic_frames(B, size, seed=2)
    bgr = np.repeat(gray[..., None], 3, axis=3)
    a = det.predict_host(gray)
    b = det.predict_host(bgr)
    np.testing.assert_array_equal(a[2], b[2])
    np.testing.assert_array_equal(a[0], b[0])
    # a coloured frame: B and R differ -> BGR->RGB order matters
    col = bgr.copy()
    col[..., 0] = 255 - col[..., 0]
    box_o, cls_o, hw = _oracle_heads(oracle, col, size)
    det.predict_host(col)
    box_g, cls_g = det.debug_head(B)
    np.testing.assert_allclose(cls_g, cls_o.numpy(), rtol=1e-3, atol=F32_LOGIT_ATOL)


def test_multiclass_head(hip_lib):
    size, B = 128, 2
    oracle, det = _models("n", size, "fp32", nc=3)
    frames, _ = fr.synthetic_frames(B, size, seed=9)
    box_o, cls_o, hw = _oracle_heads(oracle, frames, size)
    xywh, conf, anchor = det.predict_host(frames, conf=0.05)
    _, cls_g = det.debug_head(B)
    assert cls_g.shape == (B, det.anchors, 3)
    np.testing.assert_allclose(cls_g, cls_o.numpy(), rtol=1e-3, atol=F32_LOGIT_ATOL)
    _, _, anchor_o = yo.postprocess(box_o, cls_o, (size, size), hw, conf=0.05)
    np.testing.assert_array_equal(anchor, anchor_o)


def test_api_errors(hip_lib):
    _, det = _models("n", 128, "fp16", max_batch=2)
    frames, _ = fr.synthetic_frames(3, 128, seed=1)
    with pytest.raises(hip.WtkError, match="max_batch"):
        det.predict_host(frames)
    with pytest.raises(hip.WtkError, match="max_det"):
        det.predict_host(frames[:1], max_det=2)
    with pytest.raises(hip.WtkError, match="empty batch"):
        det.predict_host(frames[:0])
    w = ys.synthetic_weights("n", 1)
    w.pop("model.3")
    with pytest.raises(hip.WtkError, match="model.3"):
        hip.HipYolo(w, (128, 128), 1, width=0.25)


@pytest.mark.parametrize("dtype,B", [("fp32", 1), ("fp16", 4)])
def test_full_size_640_matches_oracle(hip_lib, dtype, B):
    """BASELINE config 2: YOLOv8s 640x640, bbox / survivor parity vs the restatement."""
    size = 640
    oracle, det = _models("s", size, dtype, max_batch=4)
    assert abs(det.macs_per_frame - 14.2158336e9) < 1e3 and det.anchors == 8400
    frames, _ = fr.synthetic_frames(B, size, seed=0)
    box_o, cls_o, hw = _oracle_heads(oracle, frames, size)
    xywh, conf, anchor = det.predict_host(frames, conf=0.1)
    box_g, cls_g = det.debug_head(B)
    err = np.abs(cls_g - cls_o.numpy()).max()
    xywh_o, _, anchor_o = yo.postprocess(box_o, cls_o, (size, size), hw, conf=0.1)
    if dtype == "fp32":
        assert err < F32_LOGIT_ATOL
        np.testing.assert_array_equal(anchor, anchor_o)
        np.testing.assert_allclose(xywh, xywh_o, rtol=0, atol=F32_BOX_ATOL)
    else:
        assert err < F16_LOGIT_ATOL
        xywh_s, _, anchor_s = yo.postprocess(torch.from_numpy(box_g), torch.from_numpy(cls_g), (size, size), hw, conf=0.1)
        np.testing.assert_array_equal(anchor, anchor_s)
