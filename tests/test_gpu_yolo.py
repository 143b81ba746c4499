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
# fp16 mode: fp16 storage of every activation; logits are O(1..10).  Bounds on the largest class- / box-logit error over all anchors of a batch,
# per model scale:
#   scale s (the benchmarked model, BASELINE configs 2-5): 0.15 — the bound this suite was written with, BEFORE any run was compared against
#     it; every measurement since lies below it (128^2 .. 1280^2: <= 0.12).  Round 3 had raised the one shared constant to 0.25 after a red
#     run (0.1834); that run was scale n, so scale s is back at its original bound.
#   scale n (width 0.25: 16-64-channel layers; not a BASELINE configuration, used by the small-shape tests): 0.25.  It is a REGRESSION bound
#     and carries no accuracy claim: with a quarter of the channels every output averages a quarter of the rounding errors of its inputs'
#     fp16 storage (error ~ 1 / sqrt(K) relative to the signal), i.e. ~ 2 x scale s's noise at equal depth -> 2 x 0.12 = 0.24, rounded up.
#     Measured: 0.184 (160^2).
# For a frame whose survivor differs from the fp32 restatement's: how far below the restatement's best logit the chosen anchor's restatement
# logit may lie (mismatch gaps <= 0.019 for the seed-0 draw these tests use, 1 792 frames at 640^2, profiles/r02_margin_study.json; other weight
# draws are noisier: tests/test_gpu_hybrid_validation.py).
F16_LOGIT_ATOL = {"s": 0.15, "n": 0.25}
F16_MISMATCH_GAP_MAX = 0.05
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


def _assert_fp16_survivors_explained(anchor, xywh, box_g, cls_g, box_o, cls_o, net_hw, hw, conf, scale="s"):
    """fp16 mode against the fp32 oracle, with NO escape for frames whose survivor differs: (1) the selection the
    GPU made from ITS OWN logits is bit-exact in index vs the oracle's selection logic on those logits; (2) the class- and
    box-logit error over ALL anchors stays below the fixed F16_LOGIT_ATOL[scale]; (3) every frame whose survivor differs from the
    fp32 oracle's is counted, and the anchor it chose must lie within the fixed F16_MISMATCH_GAP_MAX of the oracle's best logit
    (a NaN row on one side only: the oracle's best within that distance of the threshold); (4) matched survivors have
    IoU > F16_IOU_MIN.  Returns the number of mismatching frames (the rate is asserted at BASELINE scale in test_gpu_configs.py)."""
    cls_o_np, box_o_np = cls_o.numpy(), box_o.numpy()
    eps = F16_MISMATCH_GAP_MAX / 2
    measured = float(np.abs(cls_g - cls_o_np).max())
    print(f"\nfp16 logit error over all anchors: cls {measured:.4f}, box {float(np.abs(box_g - box_o_np).max()):.4f} (bound {F16_LOGIT_ATOL[scale]})")
    assert measured < F16_LOGIT_ATOL[scale] and np.abs(box_g - box_o_np).max() < F16_LOGIT_ATOL[scale]
    xywh_s, _, anchor_s = yo.postprocess(torch.from_numpy(box_g), torch.from_numpy(cls_g), net_hw, hw, conf=conf)
    np.testing.assert_array_equal(anchor, anchor_s)
    np.testing.assert_allclose(xywh, xywh_s, rtol=0, atol=2e-2)
    xywh_o, _, anchor_o = yo.postprocess(box_o, cls_o, net_hw, hw, conf=conf)
    thr = float(np.log(conf / (1 - conf)))
    best_o = cls_o_np.max(axis=2)  # [B,A]
    mismatches = 0
    for n in range(len(anchor)):
        if anchor[n] == anchor_o[n]:
            if anchor[n] >= 0:
                assert _iou_xywh(xywh[n], xywh_o[n]) > F16_IOU_MIN
            continue
        mismatches += 1
        top = best_o[n].max()
        if anchor[n] >= 0 and anchor_o[n] >= 0:
            assert top - best_o[n, anchor[n]] <= 2 * eps, (n, anchor[n], anchor_o[n], top - best_o[n, anchor[n]], eps)
        else:  # a NaN row on one side only: the oracle's best score sits within eps of the threshold
            assert abs(top - thr) <= 2 * eps, (n, top, thr, eps)
    return mismatches


@pytest.mark.parametrize("scale,size,B", [("s", 128, 2), ("s", 256, 3), ("n", 128, 3), ("n", 160, 2)])
def test_fp16_matches_oracle_within_stated_tolerance(hip_lib, scale, size, B):
    oracle, det = _models(scale, size, "fp16")
    frames, _ = fr.synthetic_frames(B, size, seed=12)
    box_o, cls_o, hw = _oracle_heads(oracle, frames, size)
    xywh, conf, anchor = det.predict_host(frames, conf=0.1)
    box_g, cls_g = det.debug_head(B)
    _assert_fp16_survivors_explained(anchor, xywh, box_g, cls_g, box_o, cls_o, (size, size), hw, 0.1, scale=scale)


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_decode_and_selection_bit_exact_on_given_logits(hip_lib, dtype):
    """Isolates box-decode / arg-max selection from conv rounding: identical logits in, survivor index
    must be identical, including ties (lowest anchor wins), the conf threshold, NaN rows, clipping."""
    size, B = 128, 8
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
    # saturated scores: the reference sorts fp32 SCORES, and every logit above 16.64 has the score 1.0f exactly -> a stable sort
    # names the lowest INDEX among them, not the largest logit (25 at anchor 150 here; round 3's kernel compared logits and failed this)
    cls[6, 311, 0], cls[6, 150, 0], cls[6, 42, 0], cls[6, 207, 0], cls[6, 41, 0] = 18.0, 25.0, 17.0, 18.0, 16.0
    # ... and one saturated anchor against ordinary ones, the last index against the first
    cls[7, A - 1, 0], cls[7, 0, 0] = 30.0, 9.0
    xywh, conf, anchor = det.decode_host(box, cls, size, size, conf=0.25)
    xywh_o, conf_o, anchor_o = yo.postprocess(torch.from_numpy(box), torch.from_numpy(cls), (size, size), (size, size), conf=0.25)
    np.testing.assert_array_equal(anchor, anchor_o)
    assert anchor[1] == -1 and np.isnan(xywh[1]).all() and anchor[2] == 37 and anchor[3] == A - 1 and anchor[4] == 0
    assert anchor[6] == 42 and conf[6] == 1.0 and anchor[7] == A - 1
    # the general NMS entry point with max_det = 1 is the same decision, row for row
    xywh_n, conf_n, _, anchor_n, cnt_n = det.decode_nms_host(box, cls, size, size, 1, conf=0.25, iou=0.7)
    np.testing.assert_array_equal(anchor_n[:, 0], anchor)
    np.testing.assert_array_equal(xywh_n[:, 0], xywh)
    np.testing.assert_array_equal(cnt_n, (anchor >= 0).astype(cnt_n.dtype))
    # the decision margin stays logit-based: frame 6's winner (logit 17) trails the largest other logit (25) by 8
    margins = det.last_margins(B)
    assert margins[6] == np.float32(17.0 - 25.0) and margins[7] == np.float32(30.0 - 9.0) and margins[2] == 0.0
    ok = anchor >= 0
    np.testing.assert_allclose(xywh[ok], xywh_o[ok], rtol=0, atol=1e-3)
    np.testing.assert_allclose(conf, conf_o, rtol=0, atol=1e-6)
    # zero box head: w = h = 15 * stride centred on the anchor (SURVEY.md §8c self-check), here stride 8
    lw = size // 8
    ax, ay = (100 % lw + 0.5) * 8, (100 // lw + 0.5) * 8
    x1, y1, x2, y2 = max(ax - 60, 0), max(ay - 60, 0), min(ax + 60, size), min(ay + 60, size)
    np.testing.assert_allclose(xywh[5], [x1, y1, x2 - x1, y2 - y1], atol=1e-3)


def _oracle_nms_rows(box, cls, net_hw, img_hw, conf, iou, max_det):
    xywh, scores = yo.decode(torch.from_numpy(box), torch.from_numpy(cls), net_hw)
    out = []
    for n in range(box.shape[0]):
        xyxy, sc, cl, idx = yo.nms(xywh[n], scores[n], conf, iou, max_det)
        b = yo.scale_boxes(net_hw, xyxy, img_hw).numpy() if len(sc) else np.zeros((0, 4), np.float32)
        out.append((np.stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], axis=1) if len(sc) else b, sc.numpy(), cl.numpy(), idx.numpy()))
    return out


@pytest.mark.parametrize("dtype,nc", [("fp32", 1), ("fp16", 1), ("fp32", 3)])
def test_general_greedy_nms_on_given_logits(hip_lib, dtype, nc):
    """SURVEY.md §8 a7 beyond the reference's max_det = 1: greedy IoU NMS on the device (wave-level arg-max + suppression sweep per kept
    box) against the restated non_max_suppression (oracle/yolo_oracle.py: nms) on identical logits — kept anchor indices in
    the same order, classes, counts; clusters of overlapping boxes, exact score ties, class-aware suppression, max_det cut-off,
    frames with no candidate."""
    size, B = 128, 5
    _, det = _models("n", size, dtype, nc=nc)
    A = det.anchors
    rng = np.random.default_rng(10 + nc)
    box = rng.normal(0, 1.0, size=(B, A, 64)).astype(np.float16).astype(np.float32)
    cls = rng.normal(-6, 1.0, size=(B, A, nc)).astype(np.float16).astype(np.float32)
    lw = size // 8
    # frame 0: three clusters of neighbouring stride-8 anchors with sharp DFL peaks (boxes overlap heavily inside a cluster)
    for cx, cy, s0 in ((3, 3, 4.0), (9, 4, 3.0), (5, 11, 2.0)):
        for dx in range(3):
            for dy in range(2):
                i = (cy + dy) * lw + cx + dx
                box[0, i] = -8.0
                box[0, i, [3, 19, 35, 51]] = 8.0  # every side ~3 bins -> 48 px boxes, neighbours shifted by 8 px: IoU ~0.7-0.8
                cls[0, i, 0] = s0 - 0.25 * (dx + 2 * dy)
    # frame 1: exact score ties between far-apart anchors -> index order
    for i in (17, 200, 90, 333):
        cls[1, i, 0] = 2.0
    cls[2] = -9.0                      # frame 2: nothing above conf
    cls[3, :40, 0] = np.linspace(3, 1, 40).astype(np.float16)   # frame 3: 40 candidates in a row (max_det cut-off)
    if nc > 1:                         # frame 4: the same box under two classes is kept twice (class-aware)
        box[4, 100] = box[4, 101] = 0.0
        cls[4, 100, 0] = 4.0
        cls[4, 101, 2] = 3.5
    else:
        cls[4, 100, 0] = 4.0
    for max_det, iou in ((300, 0.7), (5, 0.45), (1, 0.7)):
        xywh, cf, kc, an, cnt = det.decode_nms_host(box, cls, size, size, max_det, conf=0.25, iou=iou)
        want = _oracle_nms_rows(box, cls, (size, size), (size, size), 0.25, iou, max_det)
        for n in range(B):
            bw, sw, cw, iw = want[n]
            assert cnt[n] == len(iw), (n, max_det, cnt[n], len(iw))
            np.testing.assert_array_equal(an[n, : cnt[n]], iw)
            np.testing.assert_array_equal(kc[n, : cnt[n]], cw)
            np.testing.assert_allclose(cf[n, : cnt[n]], sw, rtol=0, atol=1e-6)
            np.testing.assert_allclose(xywh[n, : cnt[n]], bw, rtol=0, atol=1e-3)
            assert np.isnan(xywh[n, cnt[n] :]).all() and (an[n, cnt[n] :] == -1).all() and (kc[n, cnt[n] :] == -1).all()
        assert cnt[2] == 0
        if max_det == 300:
            assert 3 <= cnt[0] < 18 and cnt[3] > 5  # clusters collapse, a row of neighbours does not fully
            if nc > 1:
                assert {100, 101} <= set(an[4, : cnt[4]].tolist())
        if max_det == 1:  # first row == the thresholded arg-max path
            x1, c1, a1 = det.decode_host(box, cls, size, size, conf=0.25)
            np.testing.assert_array_equal(a1, an[:, 0])
            np.testing.assert_array_equal(x1, xywh[:, 0])


def test_general_nms_through_the_network(hip_lib):
    """wtk_yolo_predict_nms end to end (fp32, scale n): kept anchors / classes / counts equal to the restatement's NMS on the oracle's logits."""
    size, B, max_det = 160, 3, 20
    oracle, det = _models("n", size, "fp32", nc=2)
    frames, _ = fr.synthetic_frames(B, size, seed=17)
    box_o, cls_o, hw = _oracle_heads(oracle, frames, size)
    out = torch.full((B, max_det, 4), -1.0, device="cuda")
    cf = torch.zeros((B, max_det), device="cuda")
    kc = torch.zeros((B, max_det), dtype=torch.int32, device="cuda")
    an = torch.zeros((B, max_det), dtype=torch.int32, device="cuda")
    cnt = torch.zeros((B,), dtype=torch.int32, device="cuda")
    conf = 0.02
    det.predict_nms(torch.from_numpy(frames).cuda(), B, size, size, 1, max_det, out, cf, kc, an, cnt, conf=conf, iou=0.5,
                    stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = _oracle_nms_rows(box_o.numpy(), cls_o.numpy(), (size, size), hw, conf, 0.5, max_det)
    total = 0
    for n in range(B):
        bw, sw, cw, iw = want[n]
        k = int(cnt[n])
        assert k == len(iw)
        np.testing.assert_array_equal(an[n, :k].cpu().numpy(), iw)
        np.testing.assert_array_equal(kc[n, :k].cpu().numpy(), cw)
        np.testing.assert_allclose(out[n, :k].cpu().numpy(), bw, rtol=0, atol=F32_BOX_ATOL)
        total += k
    assert total > B  # more than one box per frame somewhere
    with pytest.raises(hip.WtkError, match="max_det"):
        det.predict_nms(torch.from_numpy(frames).cuda(), B, size, size, 1, 0, out)


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
    with pytest.raises(hip.WtkError, match="wtk_yolo_predict_nms"):  # the controller entry point stays at the reference's max_det = 1
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
        _assert_fp16_survivors_explained(anchor, xywh, box_g, cls_g, box_o, cls_o, (size, size), hw, 0.1)


_SEED_ORACLE: dict = {}


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 7])
def test_exact_modes_match_oracle_on_other_weight_draws(hip_lib, seed):
    """The detector's oracle is parity-unpinned (no ultralytics, no trained weights), so the evidence it can give is breadth: besides
    the weight draw every other test uses (seed 0), three more draws of all 63 convs at BASELINE's frame size — fp32 and f16x3 head
    logits within 2e-3 of the restatement's and the survivor index equal on every frame (the per-conv gains stored for seed 0 are a
    variance correction, not a fit: they transfer, and the test checks that both outcomes — detection and NaN row — stay reachable)."""
    size, B = 640, 4
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    frames = fr.diverse_frames(B, size, seed=30000 + 10 * seed, per_seed=1)
    oracle, det = _models("s", size, "fp32", seed=seed, max_batch=B)
    box_o, cls_o, hw = _oracle_heads(oracle, frames, size)
    xywh_o, conf_o, anchor_o = yo.postprocess(box_o, cls_o, (size, size), hw, conf=0.1)
    best = cls_o.numpy().max(axis=(1, 2))
    assert np.isfinite(best).all() and -12 < best.min() and best.max() < 12, best  # logits at unit scale: nothing saturated
    for dtype in ("fp32", "f16x3"):
        if dtype != "fp32":
            det.close()
            _, det = _models("s", size, dtype, seed=seed, max_batch=B)
        xywh, conf, anchor = det.predict_host(frames, conf=0.1)
        box_g, cls_g = det.debug_head(B)
        assert np.abs(cls_g - cls_o.numpy()).max() < F32_LOGIT_ATOL and np.abs(box_g - box_o.numpy()).max() < F32_LOGIT_ATOL, dtype
        np.testing.assert_array_equal(anchor, anchor_o)
        ok = anchor_o >= 0
        np.testing.assert_allclose(xywh[ok], xywh_o[ok], rtol=0, atol=F32_BOX_ATOL)
        np.testing.assert_allclose(conf, conf_o, rtol=0, atol=1e-4)
    det.close()


def test_widest_class_count_at_scale_s_640(hip_lib):
    """nc = 32, the widest head this library admits (wtk_yolo_create refuses more: the fused class-tower tail stores 32 padded
    channels; the reference trains with single_cls, yolo/yolo_train_config.yaml:27): every class logit within 2e-3 of the
    restatement's, survivor index equal, at BASELINE's frame size."""
    size, B, nc = 640, 2, 32
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    # The stored gains were measured for nc = 1; with 32 classes the class logits of this draw exceed 16.64, where the fp32 sigmoid is exactly
    # 1.0f for every such anchor and the reference's stable sort names the LOWEST index among them (round 3 found the device comparing logits
    # there: anchors 7254 / 7130 against the restatement's 6484).  The selection now orders by the fp32 score, so the draw is used as it is.
    w = ys.synthetic_weights("s", nc, seed=0)
    depth, width, maxch = ys.SCALES["s"]
    oracle = yo.YoloOracle(w, ys.model_dims(width, depth, maxch, nc))
    det = hip.HipYolo(w, (size, size), B, dtype="fp32", nc=nc, width=width, depth=depth, max_channels=maxch)
    frames = fr.diverse_frames(B, size, seed=31000, per_seed=1)
    box_o, cls_o, hw = _oracle_heads(oracle, frames, size)
    xywh, conf, anchor = det.predict_host(frames, conf=0.01)
    box_g, cls_g = det.debug_head(B)
    assert cls_g.shape == (B, 8400, nc)
    assert np.abs(cls_g - cls_o.numpy()).max() < F32_LOGIT_ATOL and np.abs(box_g - box_o.numpy()).max() < F32_LOGIT_ATOL
    assert cls_o.numpy().max() > 16.64, cls_o.numpy().max()  # this draw saturates the fp32 sigmoid: the case round 3 scaled away
    # selection: bit-exact against the restatement's selection logic on the SAME logits (268 800 candidates per frame, many of them at a score of
    # exactly 1.0f: lowest anchor index) ...
    xywh_s, conf_s, anchor_s = yo.postprocess(torch.from_numpy(box_g), torch.from_numpy(cls_g), (size, size), hw, conf=0.01)
    np.testing.assert_array_equal(anchor, anchor_s)
    np.testing.assert_allclose(xywh, xywh_s, rtol=0, atol=F32_BOX_ATOL)
    np.testing.assert_allclose(conf, conf_s, rtol=0, atol=1e-6)
    # ... and on its OWN logits (within 2e-3 of the device's) the restatement either names the same anchor or one whose score the device's anchor
    # equals to 1e-4 (saturated frames: both are exactly 1.0f and only the lowest saturated index can differ, by which side of 16.64 a logit fell)
    _, _, anchor_o = yo.postprocess(box_o, cls_o, (size, size), hw, conf=0.01)
    score_o = torch.sigmoid(cls_o).numpy().max(axis=2)
    for n in range(B):
        if anchor[n] != anchor_o[n]:
            assert anchor[n] >= 0 and anchor_o[n] >= 0 and score_o[n].max() - score_o[n, anchor[n]] <= 1e-4, (n, anchor[n], anchor_o[n])
    det.close()
    with pytest.raises(hip.WtkError, match="nc"):
        _models("s", size, "fp32", nc=81, max_batch=1)


@pytest.mark.parametrize("dtype", ["fp32", "f16x3"])
def test_stock_80_class_head_at_640(hip_lib, dtype):
    """A stock YOLOv8s head (nc = 80; 11.157 M parameters with Conv + BN folded) loads and runs: above 32 classes the class towers' last 1x1 is a launch of its own
    (the fused tail stores 32 couts).  Every class logit within 2e-3 of the restatement's at BASELINE's frame size, the selection bit-exact against
    the restatement's selection logic on the same logits (max over classes, fp32 score order, lowest anchor on equal scores), and the general NMS
    entry point names the restatement's classes."""
    size, B, nc = 640, 2, 80
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    w = ys.synthetic_weights("s", nc, seed=0)
    assert sum(int(np.prod(wt.shape)) + len(b) for wt, b in w.values()) == 11156528  # folded weights + biases of the 63 convs (SURVEY: 11.157 M)
    depth, width, maxch = ys.SCALES["s"]
    oracle = yo.YoloOracle(w, ys.model_dims(width, depth, maxch, nc))
    det = hip.HipYolo(w, (size, size), B, dtype=dtype, nc=nc, width=width, depth=depth, max_channels=maxch)
    frames = fr.diverse_frames(B, size, seed=32000, per_seed=1)
    box_o, cls_o, hw = _oracle_heads(oracle, frames, size)
    xywh, conf, anchor = det.predict_host(frames, conf=0.01)
    box_g, cls_g = det.debug_head(B)
    assert cls_g.shape == (B, 8400, nc)
    assert np.abs(cls_g - cls_o.numpy()).max() < F32_LOGIT_ATOL and np.abs(box_g - box_o.numpy()).max() < F32_LOGIT_ATOL
    xywh_s, conf_s, anchor_s = yo.postprocess(torch.from_numpy(box_g), torch.from_numpy(cls_g), (size, size), hw, conf=0.01)
    np.testing.assert_array_equal(anchor, anchor_s)
    np.testing.assert_allclose(xywh, xywh_s, rtol=0, atol=F32_BOX_ATOL)
    np.testing.assert_allclose(conf, conf_s, rtol=0, atol=1e-6)
    # general NMS over 80 classes (class-aware offsets): kept anchors and classes equal to the restatement's on the device's own logits
    max_det = 10
    xy, cf, kc, an, cnt = det.decode_nms_host(box_g, cls_g, size, size, max_det, conf=0.25, iou=0.7)
    want = _oracle_nms_rows(box_g, cls_g, (size, size), (size, size), 0.25, 0.7, max_det)
    for n in range(B):
        bw, sw, cw, iw = want[n]
        assert cnt[n] == len(iw)
        np.testing.assert_array_equal(an[n, : cnt[n]], iw)
        np.testing.assert_array_equal(kc[n, : cnt[n]], cw)
    det.close()


def test_letterbox_360_to_384_like_the_reference_workflow(hip_lib):
    """The reference's real shapes: 360x360 camera crops letterboxed to imgsz 384
    (initialize_experiment.ipynb cell 9): device letterbox == the restated cv2 bilinear, end to end."""
    oracle, det = _models("n", 384, "fp32", max_batch=2)
    frames, _ = fr.synthetic_frames(2, 360, seed=21)
    with torch.no_grad():
        x, hw = yo.preprocess(list(frames), 384)
        assert tuple(x.shape[2:]) == (384, 384) and hw == (360, 360)
        box_o, cls_o = oracle.forward(x)
    xywh, conf, anchor = det.predict_host(frames, conf=0.1)
    box_g, cls_g = det.debug_head(2)
    np.testing.assert_allclose(cls_g, cls_o.numpy(), rtol=1e-3, atol=F32_LOGIT_ATOL)
    xywh_o, conf_o, anchor_o = yo.postprocess(box_o, cls_o, (384, 384), hw, conf=0.1)
    np.testing.assert_array_equal(anchor, anchor_o)
    np.testing.assert_allclose(xywh, xywh_o, rtol=0, atol=F32_BOX_ATOL)
    assert (xywh[anchor >= 0, 0] + xywh[anchor >= 0, 2] <= 360 + 1e-3).all()  # boxes are in input-image pixels, clipped


def test_closed_loop_sim_with_yolo_controller_matches_oracle_controller(hip_lib, tmp_path):
    """Rows a9/a10: the controller API (on_camera_frame ring buffer, provide_movement_vector on
    deque[-pred_frame_num], _cycle_predict_all batch) driven by the harness exactly as Simulator.run does;
    integer platform moves and every logged box equal to the CPU-restatement controller's."""
    from oracle.controllers_oracle import OracleYoloController
    from wtracker_amd.controllers import HipYoloController, YoloConfig
    from wtracker_amd.sim import ExperimentConfig, TimingConfig, TrackLogger
    from harness.sim_harness import ArrayReader, Simulator

    w = ys.synthetic_weights("n", 1, seed=0)
    path = str(tmp_path / "n.wtk")
    ys.save_weights(path, w, "n", 1)
    frames, _ = fr.synthetic_frames(40, 256, seed=8)
    ec = ExperimentConfig("synthetic", 40, 60, (256, 256), 32, (128, 128))
    tc_args = (100, 40, 50, (4, 4), (0.5, 0.5))  # camera 128 px, cycle 9 frames

    def run(make):
        tc = TimingConfig(ec, *tc_args)
        ctrl = make(tc)
        moves = []
        inner = ctrl.provide_movement_vector

        def wrapped(sim):
            m = inner(sim)
            moves.append((int(m[0]), int(m[1])))
            return m

        ctrl.provide_movement_vector = wrapped
        log = TrackLogger(ctrl)
        Simulator(tc, ec, log, reader=ArrayReader(frames)).run()
        return moves, log.rows

    cfg = YoloConfig(model_path=path, device="cuda", pred_kwargs={"imgsz": 128, "conf": 0.1}, dtype="fp32", scale="n", max_batch=16)
    oracle = yo.YoloOracle(w, ys.model_dims(0.25, 0.33, 1024, 1))
    m_g, rows_g = run(lambda tc: HipYoloController(tc, cfg))
    m_o, rows_o = run(lambda tc: OracleYoloController(tc, oracle, imgsz=128, conf=0.1))
    assert m_g == m_o and len(m_g) == 4 and any(m != (0, 0) for m in m_g)
    # device-resident frames: no sim.camera_view() on the hot path, views cut + letterboxed on the device;
    # identical pixels reach the detector, so moves and logged rows are bit-identical to the host-crop controller's
    dev_frames = torch.from_numpy(frames).cuda()
    calls = []

    def make_resident(tc):
        c = HipYoloController(tc, cfg, device_frames=dev_frames)
        return c

    import harness.sim_harness as simmod
    orig_view = simmod.Simulator.camera_view
    simmod.Simulator.camera_view = lambda self: calls.append(1) or orig_view(self)
    try:
        m_d, rows_d = run(make_resident)
    finally:
        simmod.Simulator.camera_view = orig_view
    assert not calls, "the device-resident controller must never ask the simulator for a host crop"
    assert m_d == m_g and rows_d == rows_g
    # the same loop with the views entry point replaying captured hipGraphs (WTK_GRAPH_VIEWS=1, read when the handle is created: the controller
    # returns with the same device addresses per batch size, so every call after the second of a kind is a replay): identical rows
    os.environ["WTK_GRAPH_VIEWS"] = "1"
    try:
        cfg_g = YoloConfig(model_path=path, device="cuda", pred_kwargs={"imgsz": 128, "conf": 0.1}, dtype="fp32", scale="n", max_batch=16)
        m_r, rows_r = run(lambda tc: HipYoloController(tc, cfg_g, device_frames=dev_frames))
    finally:
        del os.environ["WTK_GRAPH_VIEWS"]
    assert m_r == m_g and rows_r == rows_g
    assert len(rows_g) == len(rows_o) == 36  # 4 full cycles; the trailing partial cycle is never logged
    for a, b in zip(rows_g, rows_o):
        assert (a["frame"], a["cycle"], a["phase"], a["plt_x"], a["plt_y"]) == (b["frame"], b["cycle"], b["phase"], b["plt_x"], b["plt_y"])
        np.testing.assert_allclose([a["wrm_x"], a["wrm_y"], a["wrm_w"], a["wrm_h"]], [b["wrm_x"], b["wrm_y"], b["wrm_w"], b["wrm_h"]], atol=F32_BOX_ATOL)
    # error behaviour mirrors the reference
    ctrl = HipYoloController(TimingConfig(ec, *tc_args), cfg)
    with pytest.raises(AssertionError):
        ctrl.predict([])
    cfg2 = YoloConfig(model_path=path, pred_kwargs={"imgsz": 128, "conf": 0.1, "max_det": 3}, dtype="fp32", scale="n")
    with pytest.raises(TypeError, match="max_det"):
        HipYoloController(TimingConfig(ec, *tc_args), cfg2).predict([frames[0, :128, :128]])
    none = HipYoloController(TimingConfig(ec, *tc_args), YoloConfig(model_path=path, pred_kwargs={"imgsz": 128, "conf": 0.99999}, dtype="fp32", scale="n")).predict([frames[0, :128, :128]])
    assert none.dtype == np.float64 and np.isnan(none).all()


def test_device_view_cropping_matches_view_controller(hip_lib):
    """camera views cut on the device equal ViewController.camera_view (replicate-padded
    frame, window centred on the platform position), including windows hanging over every border."""
    from harness.sim_harness import ArrayReader, ViewController

    rng = np.random.default_rng(0)
    frames = rng.integers(0, 256, size=(6, 90, 120), dtype=np.uint8)
    pos = np.array([[0, 0], [119, 89], [60, 45], [3, 80], [118, 2], [30, 30]], dtype=np.int32)
    cam = (36, 36)
    vc = ViewController(ArrayReader(frames), camera_size=cam, micro_size=(9, 9))
    expect = []
    for i in range(len(frames)):
        vc.seek(i)
        vc.set_position(int(pos[i, 0]), int(pos[i, 1]))
        expect.append(vc.camera_view())
    f_dev = torch.from_numpy(frames).cuda()
    p_dev = torch.from_numpy(pos).cuda()
    out = torch.empty((6, cam[0], cam[1]), dtype=torch.uint8, device="cuda")
    hip.crop_views(f_dev, 6, 90, 120, 1, p_dev, cam[0], cam[1], out, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), np.stack(expect))
    # colour frames + micro view
    fc = rng.integers(0, 256, size=(6, 90, 120, 3), dtype=np.uint8)
    vc = ViewController(ArrayReader(fc), camera_size=cam, micro_size=(9, 9))
    expect = []
    for i in range(len(fc)):
        vc.seek(i)
        vc.set_position(int(pos[i, 0]), int(pos[i, 1]))
        expect.append(vc.micro_view())
    out = torch.empty((6, 9, 9, 3), dtype=torch.uint8, device="cuda")
    hip.crop_views(torch.from_numpy(fc).cuda(), 6, 90, 120, 3, p_dev, 9, 9, out)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), np.stack(expect))


@pytest.mark.parametrize("frame_shape,cam,imgsz,C", [((520, 600), (360, 360), 384, 1), ((200, 260), (96, 160), 160, 3)])
def test_fused_view_crop_letterbox_matches_oracle_views(hip_lib, frame_shape, cam, imgsz, C):
    """SURVEY.md §8 f1 as written: frames + platform positions -> replicate-border camera view -> cv2-style bilinear letterbox ->
    detector, in one device pass (wtk_yolo_predict_views), at the reference's real shape (360x360 views, imgsz 384) and on a
    non-square view (rows = w, cols = h quirk of view_controller.py:171).  Against (a) the same detector fed the HOST crops of
    oracle/view_oracle.py: every head logit and every result bit-identical (the same pixels reach the stem); (b) the fp32
    restatement on those crops: logits within tolerance, survivor indices equal."""
    from oracle import view_oracle as vo

    H, W = frame_shape
    rng = np.random.default_rng(H + W + C)
    base, _ = fr.synthetic_frames(6, max(H, W), seed=31)
    frames = np.ascontiguousarray(base[:, :H, :W])
    if C == 3:
        frames = np.stack([frames, 255 - frames, rng.integers(0, 256, size=frames.shape, dtype=np.uint8)], axis=-1)
    pos = np.array([[0, 0], [W - 1, H - 1], [W // 2, H // 2], [5, H - 3], [W - 2, 7], [W // 3, H // 3]], dtype=np.int32)
    if cam[0] != cam[1]:
        # non-square views: the reference pads by (w // 2, h // 2) but slices w ROWS and h COLUMNS (view_controller.py:50-58,171),
        # so near the right / bottom border its slice runs out of the padded frame and comes back TRUNCATED (numpy slicing);
        # the device kernel replicates the border instead.  Both agree wherever the reference returns a full view.
        pos[:, 0] = np.minimum(pos[:, 0], W + 2 * (cam[0] // 2) - cam[1])
        pos[:, 1] = np.minimum(pos[:, 1], H + 2 * (cam[1] // 2) - cam[0])
    fidx = np.array([3, 1, 0, 5, 2, 2], dtype=np.int32)  # rows draw from the frame stack in any order, with repeats
    views = np.stack([vo.camera_view(frames[f], tuple(int(v) for v in p), cam) for f, p in zip(fidx, pos)])
    assert views.shape[1:3] == (cam[0], cam[1])  # rows = w, cols = h
    net = ys.letterbox_shape(cam[0], cam[1], imgsz)
    w = ys.synthetic_weights("n", 1, seed=0)
    oracle = yo.YoloOracle(w, ys.model_dims(0.25, 0.33, 1024, 1))
    det = hip.HipYolo(w, net, 8, dtype="fp32", nc=1, width=0.25, depth=0.33, max_channels=1024)
    n = len(fidx)
    out = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    cf = torch.empty((n,), dtype=torch.float32, device="cuda")
    an = torch.empty((n,), dtype=torch.int32, device="cuda")
    det.predict_views(torch.from_numpy(frames).cuda(), len(frames), H, W, C, torch.from_numpy(fidx).cuda(), torch.from_numpy(pos).cuda(), n,
                      cam[0], cam[1], out, cf, an, conf=0.1, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    box_v, cls_v = det.debug_head(n)
    xywh_h, conf_h, anchor_h = det.predict_host(views, conf=0.1)
    box_h, cls_h = det.debug_head(n)
    np.testing.assert_array_equal(box_v, box_h)
    np.testing.assert_array_equal(cls_v, cls_h)
    np.testing.assert_array_equal(out.cpu().numpy(), xywh_h)
    np.testing.assert_array_equal(an.cpu().numpy(), anchor_h)
    np.testing.assert_array_equal(cf.cpu().numpy(), conf_h)
    with torch.no_grad():
        x, hw = yo.preprocess(list(views), imgsz)
        assert tuple(x.shape[2:]) == net and hw == (cam[0], cam[1])
        box_o, cls_o = oracle.forward(x)
    np.testing.assert_allclose(cls_v, cls_o.numpy(), rtol=1e-3, atol=F32_LOGIT_ATOL)
    xywh_o, _, anchor_o = yo.postprocess(box_o, cls_o, net, hw, conf=0.1)
    np.testing.assert_array_equal(anchor_h, anchor_o)
    ok = anchor_o >= 0
    np.testing.assert_allclose(xywh_h[ok], xywh_o[ok], rtol=0, atol=F32_BOX_ATOL)
    # row n = frame n when no index is given
    det.predict_views(torch.from_numpy(frames).cuda(), len(frames), H, W, C, None, torch.from_numpy(pos).cuda(), n, cam[0], cam[1], out, cf, an,
                      conf=0.1, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    views2 = np.stack([vo.camera_view(frames[i], tuple(int(v) for v in pos[i]), cam) for i in range(n)])
    np.testing.assert_array_equal(out.cpu().numpy(), det.predict_host(views2, conf=0.1)[0])
    with pytest.raises(hip.WtkError, match="max_batch"):
        det.predict_views(torch.from_numpy(frames).cuda(), len(frames), H, W, C, None, torch.from_numpy(pos).cuda(), 9, cam[0], cam[1], out)


def test_views_call_is_ordered_behind_writes_pending_on_the_callers_stream(hip_lib, tmp_path):
    """A caller may refill `device_frames` in place between cycles: its writes sit on ITS current stream, the controller launches on a stream of its own.
    The call is ordered behind the caller's stream whenever that stream has work pending (an idle stream is not waited for: nothing to be ordered
    behind).  Here the refill is queued behind ~20 ms of matmuls on the current stream and the views call follows at once: the rows must be those of
    the NEW frames."""
    from wtracker_amd.controllers import HipYoloController, YoloConfig
    from wtracker_amd.sim import ExperimentConfig, TimingConfig

    w = ys.synthetic_weights("n", 1, seed=0)
    path = str(tmp_path / "n.wtk")
    ys.save_weights(path, w, "n", 1)
    old, _ = fr.synthetic_frames(6, 256, seed=8)
    new, _ = fr.synthetic_frames(6, 256, seed=9)
    ec = ExperimentConfig("synthetic", 6, 60, (256, 256), 32, (128, 128))
    tc = TimingConfig(ec, 100, 40, 50, (4, 4), (0.5, 0.5))
    cfg = YoloConfig(model_path=path, device="cuda", pred_kwargs={"imgsz": 128, "conf": 0.1}, dtype="fp32", scale="n", max_batch=16)
    entries = [(i, 100 + 9 * i, 120 + 5 * i, (128, 128)) for i in range(6)]
    want_new = HipYoloController(tc, cfg, device_frames=torch.from_numpy(new).cuda()).predict_views(entries)
    want_old = HipYoloController(tc, cfg, device_frames=torch.from_numpy(old).cuda()).predict_views(entries)
    assert not np.array_equal(np.nan_to_num(want_new), np.nan_to_num(want_old))
    dev = torch.from_numpy(old).cuda()
    ctrl = HipYoloController(tc, cfg, device_frames=dev)
    np.testing.assert_array_equal(ctrl.predict_views(entries), want_old)  # (buffers, streams and handles exist now: the next call starts at once)
    fresh = torch.from_numpy(new).cuda()
    a = torch.randn((4096, 4096), device="cuda")
    torch.cuda.synchronize()
    for _ in range(40):  # keeps the current stream busy for tens of milliseconds ...
        a = a @ a * 1e-3
    dev.copy_(fresh)  # ... with the refill queued behind it
    got = ctrl.predict_views(entries)
    np.testing.assert_array_equal(got, want_new)
    del a


def test_a_handle_replaced_by_a_larger_one_stays_valid_for_the_call_still_in_flight(hip_lib, tmp_path):
    """a call launched on lane 1 (the deferred cycle batch) names its detector handle in its token.  When a later, larger call makes the
    controller replace that handle (capacity 16 -> 64), the device drains first and the old handle stays readable until its tokens are collected:
    the token's rows are the rows an undisturbed call returns."""
    from wtracker_amd.controllers import HipYoloController, YoloConfig
    from wtracker_amd.sim import ExperimentConfig, TimingConfig

    w = ys.synthetic_weights("n", 1, seed=0)
    path = str(tmp_path / "n.wtk")
    ys.save_weights(path, w, "n", 1)
    frames, _ = fr.synthetic_frames(48, 256, seed=21)
    dev = torch.from_numpy(frames).cuda()
    ec = ExperimentConfig("synthetic", 48, 60, (256, 256), 32, (128, 128))
    tc = TimingConfig(ec, 100, 40, 50, (4, 4), (0.5, 0.5))
    cfg = YoloConfig(model_path=path, device="cuda", pred_kwargs={"imgsz": 128, "conf": 0.1}, dtype="fp32", scale="n", max_batch=64)
    ent = lambda n: [(i, 100 + 3 * i, 120 + 2 * i, (128, 128)) for i in range(n)]
    want15 = HipYoloController(tc, cfg, device_frames=dev).predict_views(ent(15))
    cfg.model = None
    ctrl = HipYoloController(tc, cfg, device_frames=dev)
    token = ctrl.launch_views(ent(15), lane=1)  # throughput-plan handle of 16 frames
    small = token["det"]
    got40 = ctrl.predict_views(ent(40))  # same key, larger batch: the 16-frame handle is replaced by one of 64
    assert ctrl._model.detector((128, 128), 15) is not small and small in ctrl._model._retired
    np.testing.assert_array_equal(ctrl.collect_views(token), want15)
    np.testing.assert_array_equal(got40[:15], want15)  # (batch invariance across the two handles of one plan: same kernels, same K order)


def test_converted_unfused_checkpoint_runs_on_device(hip_lib, tmp_path):
    """SURVEY.md §8 f3 end to end: a synthetic UN-FUSED ultralytics-style state dict (Conv2d + BatchNorm2d eps 1e-3 per Conv,
    plain Conv2d + bias for the Detect outputs, OIHW) -> tools/convert_ultralytics.py -> WTKYOLO1 file -> YoloConfig /
    HipYoloController on the device, against an oracle that applies conv -> BN -> SiLU explicitly on the un-fused tensors."""
    from tools import convert_ultralytics as cu
    from wtracker_amd.controllers import HipYoloController, YoloConfig
    from wtracker_amd.sim import ExperimentConfig, TimingConfig

    scale, nc, size = "n", 1, 160
    folded = ys.synthetic_weights(scale, nc, seed=0)  # well-conditioned target (gains are calibrated for seed 0); un-fold it with random BN statistics
    rng = np.random.default_rng(5)
    sd = {}
    for t in ys.conv_table(scale, nc):
        w, b = folded[t["name"]]
        w_oihw = np.ascontiguousarray(w.transpose(0, 3, 1, 2)).astype(np.float64)
        if t["act"]:
            g, var = rng.uniform(0.5, 1.5, t["cout"]), rng.uniform(0.5, 2.0, t["cout"])
            mu = rng.normal(0, 0.2, t["cout"])
            s_ = g / np.sqrt(var + cu.BN_EPS)
            sd[t["name"] + ".conv.weight"] = torch.from_numpy((w_oihw / s_[:, None, None, None]).astype(np.float32))
            sd[t["name"] + ".bn.weight"] = torch.from_numpy(g.astype(np.float32))
            sd[t["name"] + ".bn.bias"] = torch.from_numpy((b + mu * s_).astype(np.float32))
            sd[t["name"] + ".bn.running_mean"] = torch.from_numpy(mu.astype(np.float32))
            sd[t["name"] + ".bn.running_var"] = torch.from_numpy(var.astype(np.float32))
            sd[t["name"] + ".bn.num_batches_tracked"] = torch.tensor(7)
        else:
            sd[t["name"] + ".weight"] = torch.from_numpy(w_oihw.astype(np.float32))
            sd[t["name"] + ".bias"] = torch.from_numpy(b.astype(np.float32))
    sd["model.22.dfl.conv.weight"] = torch.arange(16, dtype=torch.float32).view(1, 16, 1, 1)
    ck, wtk = str(tmp_path / "unfused.pt"), str(tmp_path / "converted.wtk")
    torch.save(sd, ck)
    cu.main([ck, wtk, "--scale", scale])
    loaded, nc_file = ys.load_weights(wtk)
    assert nc_file == nc and set(loaded) == set(folded)
    frames, _ = fr.synthetic_frames(3, size, seed=14)
    ec = ExperimentConfig("synthetic", 3, 60, (size, size), 32, (size // 2, size // 2))
    ctrl = HipYoloController(TimingConfig(ec, 100, 40, 50, (4, 4), (0.5, 0.5)),
                             YoloConfig(model_path=wtk, pred_kwargs={"imgsz": size, "conf": 0.1}, scale=scale, max_batch=4))
    assert ctrl.yolo_config.dtype == "fp32"  # the reference's precision is the default
    got = ctrl.predict(list(frames))
    det = ctrl._model.detector((size, size), 3)
    box_g, cls_g = det.debug_head(3)
    oracle = yo.UnfusedYoloOracle({k: v.numpy() for k, v in sd.items()}, ys.model_dims(0.25, 0.33, 1024, nc))
    with torch.no_grad():
        x, hw = yo.preprocess(list(frames), size)
        box_o, cls_o = oracle.forward(x)
    np.testing.assert_allclose(cls_g, cls_o.numpy(), rtol=1e-3, atol=F32_LOGIT_ATOL)
    np.testing.assert_allclose(box_g, box_o.numpy(), rtol=1e-3, atol=F32_LOGIT_ATOL)
    xywh_o, _, anchor_o = yo.postprocess(box_o, cls_o, (size, size), hw, conf=0.1)
    assert (anchor_o >= 0).any()
    np.testing.assert_allclose(np.asarray(got, dtype=np.float64), np.asarray(xywh_o, dtype=np.float64), rtol=0, atol=F32_BOX_ATOL, equal_nan=True)


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_non_square_frames(hip_lib, dtype):
    """480x640-style inputs: ultralytics' auto letterbox keeps the aspect ratio (net input 96 x 160 here)."""
    H, W, imgsz = 96, 160, 160
    assert ys.letterbox_shape(H, W, imgsz) == (96, 160)
    w = ys.synthetic_weights("n", 1, seed=0)
    oracle = yo.YoloOracle(w, ys.model_dims(0.25, 0.33, 1024, 1))
    det = hip.HipYolo(w, (H, W), 3, dtype=dtype, nc=1, width=0.25, depth=0.33, max_channels=1024)
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 256, size=(3, H, W), dtype=np.uint8)
    with torch.no_grad():
        x, hw = yo.preprocess(list(frames), imgsz)
        assert tuple(x.shape[2:]) == (H, W)
        box_o, cls_o = oracle.forward(x)
    xywh, conf, anchor = det.predict_host(frames, conf=0.05)
    box_g, cls_g = det.debug_head(3)
    tol = F32_LOGIT_ATOL if dtype == "fp32" else F16_LOGIT_ATOL["n"]
    assert np.abs(cls_g - cls_o.numpy()).max() < tol and np.abs(box_g - box_o.numpy()).max() < tol
    xywh_s, _, anchor_s = yo.postprocess(torch.from_numpy(box_g), torch.from_numpy(cls_g), (H, W), hw, conf=0.05)
    np.testing.assert_array_equal(anchor, anchor_s)
    np.testing.assert_allclose(xywh, xywh_s, rtol=0, atol=2e-2)


def test_full_size_batch64_properties(hip_lib):
    """BASELINE config 3 shape (64 frames of 640x640, YOLOv8s fp16): size-independent properties —
    determinism (same input twice), batch invariance (a frame's result does not depend on its batch mates:
    sub-batches of 8 and a permuted batch give bit-identical rows)."""
    size, B = 640, 64
    _, det = _models("s", size, "fp16", max_batch=B)
    frames, _ = fr.synthetic_frames(B, size, seed=42)
    a = det.predict_host(frames, conf=0.1)
    b = det.predict_host(frames, conf=0.1)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    sub = [det.predict_host(frames[i : i + 8], conf=0.1) for i in range(0, B, 8)]
    np.testing.assert_array_equal(np.concatenate([s[0] for s in sub]), a[0])
    np.testing.assert_array_equal(np.concatenate([s[2] for s in sub]), a[2])
    perm = np.random.default_rng(0).permutation(B)
    p = det.predict_host(frames[perm], conf=0.1)
    np.testing.assert_array_equal(p[0], a[0][perm])
    np.testing.assert_array_equal(p[2], a[2][perm])
    assert (a[2] >= 0).sum() > 0 and (a[0][a[2] >= 0, 2:] > 0).all()
    # boxes lie inside the frame (scale_boxes clip)
    ok = a[2] >= 0
    assert (a[0][ok, 0] >= 0).all() and (a[0][ok, 0] + a[0][ok, 2] <= size + 1e-3).all()


@pytest.mark.parametrize("H,W,C", [(128, 128, 1), (96, 160, 3), (352, 224, 1), (640, 640, 3)])
def test_fused_kernels_equal_layer_by_layer(hip_lib, monkeypatch, H, W, C):
    """front_fused_kernel (preprocess + model.0 + model.1 + model.2.cv1), c2f32_fused_kernel (model.2's
    bottleneck + cv2) and the fused 1x1 tail of the Detect box tower round to fp16 where the layer-by-layer kernels store fp16 and walk K in the same order:
    every head logit must be bit-identical, on full tiles, ragged tiles (maps not a multiple of 16), gray and
    BGR frames, with either or both fusions active."""
    B = 3
    w = ys.synthetic_weights("s", 1, seed=0)
    depth, width, maxch = ys.SCALES["s"]
    rng = np.random.default_rng(H * 7 + W + C)
    frames = rng.integers(0, 256, size=(B, H, W) if C == 1 else (B, H, W, 3), dtype=np.uint8)
    outs = []
    for no_front, no_c2f, no_tail, no_wide in (("1", "1", "1", "1"), ("0", "1", "1", "1"), ("1", "0", "1", "1"), ("0", "0", "1", "1"), ("0", "0", "0", "1"),
                                               ("1", "1", "1", "0"), ("0", "0", "0", "0")):
        monkeypatch.setenv("WTK_NO_FUSED_FRONT", no_front)
        monkeypatch.setenv("WTK_NO_FUSED_C2F", no_c2f)
        monkeypatch.setenv("WTK_NO_FUSED_TAIL", no_tail)  # Detect box tower: last 1x1 inside the 3x3's epilogue
        monkeypatch.setenv("WTK_NO_WIDE_1X1", no_wide)    # 1x1 convs: 256 x 128 tile / three-stage ring instead of conv_igemm_kernel
        monkeypatch.setenv("WTK_NO_IGEMM_TAIL", no_wide)  # model.4.cv1 inside the epilogue of model.3 (same switch position as the wide kernel)
        det = hip.HipYolo(w, (H, W), B, dtype="fp16", nc=1, width=width, depth=depth, max_channels=maxch)
        res = det.predict_host(frames, conf=0.05)
        outs.append((res, det.debug_head(B), det.debug_tensor(3, B)))  # conv 3 = model.2.cv2
        del det
    ref_res, (ref_box, ref_cls), ref_t2 = outs[0]
    for res, (box, cls), t2 in outs[1:]:
        np.testing.assert_array_equal(t2, ref_t2)
        np.testing.assert_array_equal(box, ref_box)
        np.testing.assert_array_equal(cls, ref_cls)
        for x, y in zip(res, ref_res):
            np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("dtype,H,W", [("fp16", 128, 128), ("fp32", 96, 160), ("fp16", 352, 224)])
def test_two_source_upsample_loader_equals_materialised_concat(hip_lib, monkeypatch, dtype, H, W):
    """nn.Upsample(2x nearest) + Concat in the FPN: the consumer's 1x1 conv reads the half-resolution tensor at
    (y/2, x/2) instead of a materialised 4x copy.  Same values in the same K order: bit-identical logits."""
    B = 2
    w = ys.synthetic_weights("s", 1, seed=1)
    depth, width, maxch = ys.SCALES["s"]
    frames = np.random.default_rng(H + W).integers(0, 256, size=(B, H, W), dtype=np.uint8)
    outs = []
    for mat in ("1", "0"):
        monkeypatch.setenv("WTK_MATERIALIZE_UPSAMPLE", mat)
        det = hip.HipYolo(w, (H, W), B, dtype=dtype, nc=1, width=width, depth=depth, max_channels=maxch)
        res = det.predict_host(frames, conf=0.05)
        outs.append((res, det.debug_head(B)))
        del det
    (ra, (box_a, cls_a)), (rb, (box_b, cls_b)) = outs
    np.testing.assert_array_equal(box_a, box_b)
    np.testing.assert_array_equal(cls_a, cls_b)
    for x, y in zip(ra, rb):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("dtype,B,H,W", [("fp16", 5, 352, 224), ("fp32", 5, 352, 224), ("fp16", 64, 640, 640)])
def test_three_slab_counted_wait_schedule_equals_two_slab(hip_lib, monkeypatch, dtype, B, H, W):
    """conv3x3_halo_kernel with three weight slabs and counted vmcnt waits (LDS-DMA in flight across the tap barrier),
    one tile per block and persistent, against the two-slab / vmcnt(0) schedule: same arithmetic, so bit-identical
    logits; several runs, because a wait placed one tap too late would show up as run-to-run differences.  The
    persistent form is only dispatched at >= 1.5 tiles per CU: the 64 x 640^2 case (BASELINE shape) exercises it on
    the stride-8 and stride-16 maps."""
    w = ys.synthetic_weights("s", 1, seed=2)
    depth, width, maxch = ys.SCALES["s"]
    frames = np.random.default_rng(9).integers(0, 256, size=(B, H, W), dtype=np.uint8)
    outs = []
    for slabs, persist, small in (("2", "0", "0"), ("3", "0", "0"), ("3", "1", "0"), ("3", "1", "1"), ("3", "1", "1")):
        monkeypatch.setenv("WTK_HALO_SLABS", slabs)
        monkeypatch.setenv("WTK_HALO_PERSIST", persist)  # persistent form: the tap pipeline runs on across tiles
        monkeypatch.setenv("WTK_HALO_SMALL_BLOCKS", small)  # 128-pixel blocks on maps that would leave CUs idle
        det = hip.HipYolo(w, (H, W), B, dtype=dtype, nc=1, width=width, depth=depth, max_channels=maxch)
        res = det.predict_host(frames, conf=0.05)
        outs.append((res, det.debug_head(min(B, 8))))
        del det
    ref_res, (ref_box, ref_cls) = outs[0]
    for res, (box, cls) in outs[1:]:
        np.testing.assert_array_equal(box, ref_box)
        np.testing.assert_array_equal(cls, ref_cls)
        for x, y in zip(res, ref_res):
            np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("B,H,W", [(64, 640, 640), (24, 1280, 736), (33, 640, 512)])
def test_weight_stationary_64ch_kernel_equals_window_kernel(hip_lib, monkeypatch, B, H, W):
    """conv3x3_ws64_kernel (all nine 64x64 tap slabs resident in LDS, two wave groups alternating multiply / stage + epilogue, one
    barrier per tile) against conv3x3_halo_kernel on the same layers (model.4 / model.15 bottlenecks; WTK_NO_WS64=1): same fragment
    layouts, K order, bias-initialised accumulators and SiLU, so every head logit and result is bit-identical — on the BASELINE
    shape, on a 3-strip geometry (160 x 92 maps), and with a batch that leaves the two groups of a block unequal tile counts;
    two runs of the default build catch a missing wait as a run-to-run difference."""
    w = ys.synthetic_weights("s", 1, seed=5)
    depth, width, maxch = ys.SCALES["s"]
    frames = np.random.default_rng(B + H).integers(0, 256, size=(B, H, W), dtype=np.uint8)
    outs = []
    for off in ("1", "0", "0"):
        monkeypatch.setenv("WTK_NO_WS64", off)
        det = hip.HipYolo(w, (H, W), B, dtype="fp16", nc=1, width=width, depth=depth, max_channels=maxch)
        res = det.predict_host(frames, conf=0.05)
        idx = [i for i, t in enumerate(ys.conv_table("s", 1)) if t["name"] == "model.4.m.1.cv2"][0]
        outs.append((res, det.debug_head(min(B, 8)), det.debug_tensor(idx, min(B, 4))))
        det.close()
    ref_res, (ref_box, ref_cls), ref_t = outs[0]
    for res, (box, cls), t in outs[1:]:
        np.testing.assert_array_equal(t, ref_t)
        np.testing.assert_array_equal(box, ref_box)
        np.testing.assert_array_equal(cls, ref_cls)
        for x, y in zip(res, ref_res):
            np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("B,H,W", [(64, 640, 640), (3, 352, 224), (9, 1280, 736)])
def test_stride2_window_kernel_matches_implicit_gemm(hip_lib, monkeypatch, B, H, W):
    """conv3x3_s2_kernel (strided 3x3 convs through an LDS-resident window of the four input parity planes) against the implicit-GEMM
    path (WTK_NO_S2WIN=1).  The window kernel walks K chunk-major / plane-major, the implicit GEMM tap-major: the same products in
    another fp32 summation order, so outputs agree to rounding, not bit for bit — model.5 / 7 / 16 / 19 outputs within two fp16 ulps of
    their scale, head logits within 1 % of the logit scale, survivors equal except on near-ties; two default runs must be bit-identical (a missing
    wait shows as a run-to-run difference).  Shapes: BASELINE (256- and 128-pixel blocks), ragged small maps, non-square 1280."""
    w = ys.synthetic_weights("s", 1, seed=0)  # the calibrated seed: activations and logits stay O(1 .. 10)
    depth, width, maxch = ys.SCALES["s"]
    frames = np.random.default_rng(B * H + W).integers(0, 256, size=(B, H, W), dtype=np.uint8)
    table = ys.conv_table("s", 1)
    idx = {nm: [i for i, t in enumerate(table) if t["name"] == nm][0] for nm in ("model.5", "model.7", "model.16", "model.19")}
    outs = []
    for off in ("1", "0", "0"):
        monkeypatch.setenv("WTK_NO_S2WIN", off)
        det = hip.HipYolo(w, (H, W), B, dtype="fp16", nc=1, width=width, depth=depth, max_channels=maxch)
        res = det.predict_host(frames, conf=0.05)
        nb = min(B, 4)
        outs.append((res, det.debug_head(nb), {nm: det.debug_tensor(i, nb) for nm, i in idx.items()}))
        det.close()
    (res_r, (box_r, cls_r), t_r), (res_a, (box_a, cls_a), t_a), (res_b, (box_b, cls_b), t_b) = outs
    for x, y in zip(res_a, res_b):  # determinism of the window kernel
        np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(cls_a, cls_b)
    for nm in idx:
        scale = max(1.0, float(np.abs(t_r[nm]).max()))
        err = float(np.abs(t_a[nm] - t_r[nm]).max())
        assert err <= 2.0 * scale * 2.0 ** -10, (nm, err, scale)  # two fp16 ulps at the tensor's scale (inputs of later layers differ by an ulp already)
    lscale = max(1.0, float(np.abs(cls_r).max()), float(np.abs(box_r).max()))
    assert np.abs(cls_a - cls_r).max() < 0.01 * lscale and np.abs(box_a - box_r).max() < 0.01 * lscale
    same = res_a[2] == res_r[2]
    assert same.mean() >= 0.9
    np.testing.assert_allclose(res_a[0][same & (res_r[2] >= 0)], res_r[0][same & (res_r[2] >= 0)], rtol=0, atol=0.05)


@pytest.mark.parametrize("B,H,W", [(8, 640, 640), (3, 352, 224), (5, 1280, 736)])
def test_split_32ch_window_kernel_matches_implicit_gemm(hip_lib, monkeypatch, B, H, W):
    """conv3x3_c32_split_kernel (f16x3 handles: model.2's 32 -> 32 channel 3x3 layers with the window and all nine weight slabs in LDS,
    one barrier per block) against the implicit-GEMM path it replaces (WTK_NO_C32S=1).  Both sum the same split products (hi.hi apart from
    the 2^-11 cross terms) in fp32, in different orders: the two layers' outputs agree to a few fp32 ulps of their scale, the head logits
    to 1e-4 of theirs, and the survivors are the same rows; two default runs are bit-identical.  Shapes: four 40-column strips (BASELINE
    map), ragged strips on a small map (88 x 56), eight strips on a non-square map."""
    w = ys.synthetic_weights("s", 1, seed=0)
    depth, width, maxch = ys.SCALES["s"]
    frames = np.random.default_rng(B * H + W).integers(0, 256, size=(B, H, W), dtype=np.uint8)
    table = ys.conv_table("s", 1)
    idx = {nm: [i for i, t in enumerate(table) if t["name"] == nm][0] for nm in ("model.2.m.0.cv1", "model.2.m.0.cv2", "model.2.cv2")}
    outs = []
    for off in ("1", "0", "0"):
        monkeypatch.setenv("WTK_NO_C32S", off)
        det = hip.HipYolo(w, (H, W), B, dtype="f16x3", nc=1, width=width, depth=depth, max_channels=maxch)
        res = det.predict_host(frames, conf=0.05)
        nb = min(B, 4)
        outs.append((res, det.debug_head(nb), {nm: det.debug_tensor(i, nb) for nm, i in idx.items()}))
        det.close()
    (res_r, (box_r, cls_r), t_r), (res_a, (box_a, cls_a), t_a), (res_b, (box_b, cls_b), t_b) = outs
    for x, y in zip(res_a, res_b):
        np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(cls_a, cls_b)
    for nm in idx:
        np.testing.assert_array_equal(t_a[nm], t_b[nm])
        scale = max(1.0, float(np.abs(t_r[nm]).max()))
        err = float(np.abs(t_a[nm] - t_r[nm]).max())
        assert err <= 16 * scale * 2.0 ** -23, (nm, err, scale)
    lscale = max(1.0, float(np.abs(cls_r).max()), float(np.abs(box_r).max()))
    assert np.abs(cls_a - cls_r).max() < 1e-4 * lscale and np.abs(box_a - box_r).max() < 1e-4 * lscale
    np.testing.assert_array_equal(res_a[2], res_r[2])
    det_rows = res_r[2] >= 0
    np.testing.assert_allclose(res_a[0][det_rows], res_r[0][det_rows], rtol=0, atol=1e-3)


@pytest.mark.parametrize("H,W,C", [(128, 128, 1), (96, 160, 3)])
def test_fused_front_and_c2f_tail_match_oracle_layers(hip_lib, H, W, C):
    """Layer-level parity of the two fused kernels against the CPU restatement (not only through the head logits):
    model.2.cv1's output (= front_fused_kernel's result) and model.2's output (= c2f32_fused_kernel's result) in
    fp16 mode; activations there are O(1), the tolerance is a few fp16 ulps accumulated over 3 / 6 layers."""
    B = 2
    w = ys.synthetic_weights("s", 1, seed=4)
    depth, width, maxch = ys.SCALES["s"]
    oracle = yo.YoloOracle(w, ys.model_dims(width, depth, maxch, 1))
    rng = np.random.default_rng(H + 3 * W + C)
    frames = rng.integers(0, 256, size=(B, H, W) if C == 1 else (B, H, W, 3), dtype=np.uint8)
    det = hip.HipYolo(w, (H, W), B, dtype="fp16", nc=1, width=width, depth=depth, max_channels=maxch)
    det.predict_host(frames, conf=0.05)
    cv1_g = det.debug_tensor(2, B)  # [B,h,w,64]
    c2f_g = det.debug_tensor(3, B)
    with torch.no_grad():
        x, _ = yo.preprocess(list(frames), max(H, W))
        assert tuple(x.shape[2:]) == (H, W)
        x1 = oracle.conv("model.1", oracle.conv("model.0", x, 2), 2)
        cv1_o = oracle.conv("model.2.cv1", x1).permute(0, 2, 3, 1).numpy()
        c2f_o = oracle.c2f("model.2", x1, 1, True).permute(0, 2, 3, 1).numpy()
    assert cv1_g.shape == cv1_o.shape and c2f_g.shape == c2f_o.shape
    scale = max(1.0, float(np.abs(cv1_o).max()))
    assert np.abs(cv1_g - cv1_o).max() < 2e-2 * scale
    assert np.abs(c2f_g - c2f_o).max() < 4e-2 * max(1.0, float(np.abs(c2f_o).max()))


@pytest.mark.parametrize("dtype,scale", [("fp16", "s"), ("fp32", "s"), ("fp16", "n")])
def test_kernel_profile_accounts_for_every_conv_mac(hip_lib, dtype, scale):
    """bench.py's roofline object is built from wtk_yolo_get_kernel_profile: the FLOPs it attributes to the kernels must add
    up to the algorithmic work of the forward pass (2 x macs_per_frame x B), whatever mix of fused / plain kernels ran, and
    the public 'conv' class must be the sum of the MFMA conv kernels."""
    size, B, steps = 128, 3, 2
    _, det = _models(scale, size, dtype)
    frames, _ = fr.synthetic_frames(B, size, seed=5)
    res_plain = det.predict_host(frames, conf=0.1)
    det.set_profiling(True)
    for _ in range(steps):
        res_prof = det.predict_host(frames, conf=0.1)
    kp = det.get_kernel_profile()
    cp = det.get_profile()
    det.set_profiling(False)
    for a, b in zip(res_plain, res_prof):  # profiling (one stream, event brackets) does not change results
        np.testing.assert_array_equal(a, b)
    total = sum(k["flops"] for k in kp.values())
    assert total == pytest.approx(2.0 * det.macs_per_frame * B * steps, rel=1e-9)
    conv_ids = ["conv_igemm_kernel+conv1x1_wide_kernel", "conv3x3_halo_kernel", "front_fused_kernel+c2f32_fused_kernel", "conv3x3_c32_kernel"]
    assert cp["conv"]["launches"] == sum(kp[k]["launches"] for k in conv_ids)
    assert cp["conv"]["total_ms"] == pytest.approx(sum(kp[k]["total_ms"] for k in conv_ids), rel=1e-9)
    assert all(k["total_ms"] > 0 for k in kp.values() if k["launches"])
    det.close()
