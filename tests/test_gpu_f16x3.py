"""GPU parity of the split-fp16 mode (dtype "f16x3": hi + lo * 2^-11 pairs, three fp16 MFMAs per product) — held to the
fp32 mode's tolerances against the CPU restatement (oracle/yolo_oracle.py), and compared tensor by tensor with the exact-fp32
MFMA mode of the same library.  Everything goes through the C ABI."""
import numpy as np
import pytest
import torch

from oracle import yolo_oracle as yo
from wtracker_amd import frames as fr
from wtracker_amd import hip
from wtracker_amd import yolo_spec as ys

pytestmark = pytest.mark.gpu

F32_LOGIT_ATOL = 2e-3  # the fp32 mode's bounds (tests/test_gpu_yolo.py)
F32_BOX_ATOL = 2e-2


def _det(scale, hw, dtype, max_batch, nc=1, seed=0):
    w = ys.synthetic_weights(scale, nc, seed=seed)
    depth, width, maxch = ys.SCALES[scale]
    det = hip.HipYolo(w, hw, max_batch, dtype=dtype, nc=nc, width=width, depth=depth, max_channels=maxch)
    return w, det, ys.model_dims(width, depth, maxch, nc)


@pytest.mark.parametrize("size,B", [(128, 2), (256, 3), (352, 2)])
def test_f16x3_head_logits_and_boxes_match_oracle(hip_lib, size, B):
    w, det, dims = _det("s", (size, size), "f16x3", B)
    oracle = yo.YoloOracle(w, dims)
    frames, _ = fr.synthetic_frames(B, size, seed=11)
    with torch.no_grad():
        x, hw = yo.preprocess(list(frames), size)
        box_o, cls_o = oracle.forward(x)
    xywh, conf, anchor = det.predict_host(frames, conf=0.1)
    box_g, cls_g = det.debug_head(B)
    np.testing.assert_allclose(cls_g, cls_o.numpy(), rtol=1e-3, atol=F32_LOGIT_ATOL)
    np.testing.assert_allclose(box_g, box_o.numpy(), rtol=1e-3, atol=F32_LOGIT_ATOL)
    xywh_o, conf_o, anchor_o = yo.postprocess(box_o, cls_o, (size, size), hw, conf=0.1)
    np.testing.assert_array_equal(anchor, anchor_o)
    np.testing.assert_allclose(xywh, xywh_o, rtol=0, atol=F32_BOX_ATOL)
    np.testing.assert_allclose(conf, conf_o, rtol=0, atol=1e-4)


@pytest.mark.parametrize("H,W,C,no_halo", [(224, 160, 1, "0"), (128, 192, 3, "0"), (224, 160, 1, "1")])
def test_f16x3_every_tensor_against_the_fp32_mode(hip_lib, monkeypatch, H, W, C, no_halo):
    """Every conv output of the split mode against the exact-fp32 MFMA mode of the same library on the same frames: a split
    product carries the error of an fp32 product (2^-22 relative per operand pair), so the tensors agree to fp32 rounding
    accumulated over the layers in front of them — three orders of magnitude below the fp16 mode's 2^-11 per layer."""
    B = 3
    monkeypatch.setenv("WTK_FRONT_DEBUG", "1")  # the fused front (front_fused_split_kernel) also writes the model.0 / model.1 tensors it keeps in LDS
    monkeypatch.setenv("WTK_NO_FUSED_TAIL", "1")  # the box towers' 3x3 outputs exist as tensors only when their 1x1 is a launch of its own
    monkeypatch.setenv("WTK_NO_HALO", no_halo)  # "1": every 3x3 conv through the split implicit GEMM instead of the split window kernels
    monkeypatch.setenv("WTK_NO_S2WIN", no_halo)
    rng = np.random.default_rng(H + W)
    frames = rng.integers(0, 256, size=(B, H, W) if C == 1 else (B, H, W, 3), dtype=np.uint8)
    outs = {}
    for dtype in ("fp32", "f16x3"):
        _, det, _ = _det("s", (H, W), dtype, B)
        res = det.predict_host(frames, conf=0.05)
        tensors = {}
        for i, t in enumerate(ys.conv_table("s", 1)):
            try:
                tensors[t["name"]] = det.debug_tensor(i, B)
            except hip.WtkError:
                pass  # a conv computed inside another op (concatenated Detect stems report under their first blob)
        outs[dtype] = (res, det.debug_head(B), tensors)
        det.close()
    (res_r, (box_r, cls_r), t_r), (res_s, (box_s, cls_s), t_s) = outs["fp32"], outs["f16x3"]
    assert set(t_r) == set(t_s) and len(t_r) >= 50
    worst = 0.0
    for nm in t_r:
        scale = max(1.0, float(np.abs(t_r[nm]).max()))
        err = float(np.abs(t_s[nm] - t_r[nm]).max()) / scale
        worst = max(worst, err)
        assert err < 2e-5, (nm, err, scale)
    print(f"f16x3 vs fp32 mode, {len(t_r)} tensors: worst scaled difference {worst:.2e}")
    lscale = max(1.0, float(np.abs(cls_r).max()), float(np.abs(box_r).max()))
    assert np.abs(cls_s - cls_r).max() < 2e-5 * lscale and np.abs(box_s - box_r).max() < 2e-5 * lscale
    np.testing.assert_array_equal(res_s[2], res_r[2])
    np.testing.assert_allclose(res_s[0], res_r[0], rtol=0, atol=2e-3, equal_nan=True)


@pytest.mark.parametrize("B,H,W,C", [(3, 128, 128, 1), (3, 96, 160, 3), (3, 352, 224, 1), (4, 640, 640, 3), (2, 1280, 736, 1)])
def test_f16x3_fused_front_equals_layer_by_layer(hip_lib, monkeypatch, B, H, W, C):
    """front_fused_split_kernel (preprocess + model.0 + model.1 + model.2.cv1 of an f16x3 handle in one persistent kernel, intermediates
    as split rows in LDS) against the three stand-alone launches (WTK_NO_FUSED_FRONT=1): the same operands, K order and instruction
    sequence per stage, so model.0, model.1 (test hook WTK_FRONT_DEBUG=1), model.2.cv1, every head logit and every result are
    bit-identical — full tiles, ragged tiles (maps that are no multiple of the 16 x 4 tile), gray and BGR frames; two fused runs
    catch a missing wait as a run-to-run difference."""
    w = ys.synthetic_weights("s", 1, seed=0)
    depth, width, maxch = ys.SCALES["s"]
    rng = np.random.default_rng(H * 7 + W + C)
    frames = rng.integers(0, 256, size=(B, H, W) if C == 1 else (B, H, W, 3), dtype=np.uint8)
    monkeypatch.setenv("WTK_FRONT_DEBUG", "1")
    outs = []
    for off in ("1", "0", "0"):
        monkeypatch.setenv("WTK_NO_FUSED_FRONT", off)
        det = hip.HipYolo(w, (H, W), B, dtype="f16x3", nc=1, width=width, depth=depth, max_channels=maxch)
        res = det.predict_host(frames, conf=0.05)
        outs.append((res, det.debug_head(B), [det.debug_tensor(i, B) for i in (0, 1, 2, 3)]))  # model.0, model.1, model.2.cv1, model.2.cv2
        det.close()
    ref_res, (ref_box, ref_cls), ref_t = outs[0]
    assert all(np.abs(t).max() > 0 for t in ref_t)
    for res, (box, cls), ts in outs[1:]:
        for t, r in zip(ts, ref_t):
            np.testing.assert_array_equal(t, r)
        np.testing.assert_array_equal(box, ref_box)
        np.testing.assert_array_equal(cls, ref_cls)
        for x, y in zip(res, ref_res):
            np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("B,H,W", [(3, 128, 128), (2, 352, 224), (4, 640, 640), (2, 1280, 736)])
def test_f16x3_fused_box_tail_equals_two_launches(hip_lib, monkeypatch, B, H, W):
    """The Detect box towers' last 1x1 (64 -> 64, fp32 logits out) inside the epilogue of the split 3x3 before it (split rows in LDS, wave-local)
    against the two launches (WTK_NO_FUSED_TAIL=1): the same split products in the same order — every head logit and result bit-identical, at the
    three levels' block sizes (256- and 128-pixel blocks) and on ragged maps; two fused runs catch a missing wait."""
    w = ys.synthetic_weights("s", 1, seed=0)
    depth, width, maxch = ys.SCALES["s"]
    frames = np.random.default_rng(H + 3 * W).integers(0, 256, size=(B, H, W), dtype=np.uint8)
    outs = []
    for off in ("1", "0", "0"):
        monkeypatch.setenv("WTK_NO_FUSED_TAIL", off)
        det = hip.HipYolo(w, (H, W), B, dtype="f16x3", nc=1, width=width, depth=depth, max_channels=maxch)
        res = det.predict_host(frames, conf=0.05)
        outs.append((res, det.debug_head(B)))
        det.close()
    ref_res, (ref_box, ref_cls) = outs[0]
    assert np.abs(ref_box).max() > 0
    for res, (box, cls) in outs[1:]:
        np.testing.assert_array_equal(box, ref_box)
        np.testing.assert_array_equal(cls, ref_cls)
        for x, y in zip(res, ref_res):
            np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("B,H,W,nc", [(3, 128, 128, 1), (2, 352, 224, 3), (4, 640, 640, 1), (1, 1280, 736, 20)])
def test_f16x3_fused_class_tail_equals_two_launches(hip_lib, monkeypatch, B, H, W, nc):
    """Round 4: the Detect class towers' last 1x1 (128 -> nc, stored as up to 32 couts, fp32 logits) inside the epilogue of the split 3x3 before it
    (`TAIL + SPLIT` instantiation at the 128-cout tile: split rows of both cout-waves in LDS, one barrier, K order of the stand-alone split 1x1)
    against the launch of its own (WTK_NO_SPLIT_CLS_TAIL=1; the box towers stay fused in both): every class and box logit and every result
    bit-identical, on 256- and 128-pixel blocks, ragged maps and several class counts; two fused runs catch a missing wait."""
    w = ys.synthetic_weights("s", nc, seed=0)
    depth, width, maxch = ys.SCALES["s"]
    frames = np.random.default_rng(H + 5 * W + nc).integers(0, 256, size=(B, H, W), dtype=np.uint8)
    outs = []
    for off in ("1", "0", "0"):
        monkeypatch.setenv("WTK_NO_SPLIT_CLS_TAIL", off)
        det = hip.HipYolo(w, (H, W), B, dtype="f16x3", nc=nc, width=width, depth=depth, max_channels=maxch)
        res = det.predict_host(frames, conf=0.01)
        outs.append((res, det.debug_head(B)))
        det.close()
    ref_res, (ref_box, ref_cls) = outs[0]
    assert np.abs(ref_cls).max() > 0 and ref_cls.shape[2] == nc
    for res, (box, cls) in outs[1:]:
        np.testing.assert_array_equal(cls, ref_cls)
        np.testing.assert_array_equal(box, ref_box)
        for x, y in zip(res, ref_res):
            np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("size,B", [(640, 4), (1280, 2)])
def test_f16x3_full_size_survivors_equal_oracle(hip_lib, size, B):
    """BASELINE configs 2 and 5 frame shapes (1280x1280: the window kernels cut the 160-column maps into two strips)."""
    w, det, dims = _det("s", (size, size), "f16x3", B)
    oracle = yo.YoloOracle(w, dims)
    frames, _ = fr.synthetic_frames(B, size, seed=0)
    with torch.no_grad():
        x, hw = yo.preprocess(list(frames), size)
        box_o, cls_o = oracle.forward(x)
    xywh, conf, anchor = det.predict_host(frames, conf=0.1)
    box_g, cls_g = det.debug_head(B)
    assert np.abs(cls_g - cls_o.numpy()).max() < F32_LOGIT_ATOL
    xywh_o, _, anchor_o = yo.postprocess(box_o, cls_o, (size, size), hw, conf=0.1)
    np.testing.assert_array_equal(anchor, anchor_o)
    np.testing.assert_allclose(xywh, xywh_o, rtol=0, atol=F32_BOX_ATOL)


def test_f16x3_rejects_scales_it_cannot_split(hip_lib):
    w = ys.synthetic_weights("n", 1, seed=0)
    depth, width, maxch = ys.SCALES["n"]
    with pytest.raises(hip.WtkError, match="WTK_F16X3"):
        hip.HipYolo(w, (128, 128), 1, dtype="f16x3", nc=1, width=width, depth=depth, max_channels=maxch)


def test_closed_loop_controller_in_auto_precision_equals_oracle_controller(hip_lib, tmp_path):
    """YoloConfig(dtype="auto") at scale s resolves to the split mode; the closed loop (camera views depend on the previous cycle's
    movement) gives the integer platform moves and logged boxes of the CPU-restatement controller — host crops and device-resident frames."""
    from oracle.controllers_oracle import OracleYoloController
    from wtracker_amd.controllers import HipYoloController, YoloConfig
    from wtracker_amd.sim import ExperimentConfig, TimingConfig, TrackLogger
    from harness.sim_harness import ArrayReader, Simulator

    w = ys.synthetic_weights("s", 1, seed=0)
    path = str(tmp_path / "s.wtk")
    ys.save_weights(path, w, "s", 1)
    depth, width, maxch = ys.SCALES["s"]
    frames, _ = fr.synthetic_frames(40, 256, seed=8)
    ec = ExperimentConfig("synthetic", 40, 60, (256, 256), 32, (128, 128))
    tc_args = (100, 40, 50, (4, 4), (0.5, 0.5))

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
        return moves, log.rows, ctrl

    cfg = YoloConfig(model_path=path, device="cuda", pred_kwargs={"imgsz": 128, "conf": 0.1}, dtype="auto", scale="s", max_batch=16)
    oracle = yo.YoloOracle(w, ys.model_dims(width, depth, maxch, 1))
    m_g, rows_g, ctrl = run(lambda tc: HipYoloController(tc, cfg))
    assert [d.dtype for d in ctrl._model._dets.values()] == ["f16x3"]
    m_o, rows_o, _ = run(lambda tc: OracleYoloController(tc, oracle, imgsz=128, conf=0.1))
    assert m_g == m_o and len(m_g) == 4
    m_d, rows_d, _ = run(lambda tc: HipYoloController(tc, cfg, device_frames=torch.from_numpy(frames).cuda()))
    assert m_d == m_g and rows_d == rows_g
    assert len(rows_g) == len(rows_o) == 36
    for a, b in zip(rows_g, rows_o):
        assert (a["frame"], a["cycle"], a["phase"], a["plt_x"], a["plt_y"]) == (b["frame"], b["cycle"], b["phase"], b["plt_x"], b["plt_y"])
        np.testing.assert_allclose([a["wrm_x"], a["wrm_y"], a["wrm_w"], a["wrm_h"]], [b["wrm_x"], b["wrm_y"], b["wrm_w"], b["wrm_h"]], atol=F32_BOX_ATOL)


@pytest.mark.parametrize("dtype,size,B,n", [("f16x3", 256, 8, 3), ("f16x3", 640, 6, 1), ("fp16", 256, 8, 5), ("fp32", 160, 6, 2), ("f16x3", 128, 4, 0)])
def test_dynamic_batch_rows_equal_the_static_run(hip_lib, dtype, size, B, n):
    """wtk_yolo_set_dynamic_batch: the handle reads the number of batch rows that matter from device memory; kernels skip the tiles of
    the images behind it.  The first n rows must be bit-identical to a static run of the same batch (tiles straddling the limit are
    computed whole; rows behind it are scratch)."""
    w, det, _ = _det("s", (size, size), dtype, B)
    frames, _ = fr.synthetic_frames(B, size, seed=5)
    dev = torch.from_numpy(frames).cuda()
    out = [torch.empty((B, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
    cf = [torch.empty((B,), dtype=torch.float32, device="cuda") for _ in range(2)]
    an = [torch.empty((B,), dtype=torch.int32, device="cuda") for _ in range(2)]
    det.predict(dev, B, size, size, 1, out[0], cf[0], an[0], conf=0.1)
    torch.cuda.synchronize()
    box_s, cls_s = det.debug_head(B)
    n_dev = torch.tensor([n], dtype=torch.int32, device="cuda")
    det.set_dynamic_batch(n_dev)
    det.predict(dev, B, size, size, 1, out[1], cf[1], an[1], conf=0.1)
    torch.cuda.synchronize()
    box_d, cls_d = det.debug_head(B)
    np.testing.assert_array_equal(cls_d[:n], cls_s[:n])
    np.testing.assert_array_equal(box_d[:n], box_s[:n])
    np.testing.assert_array_equal(out[1][:n].cpu().numpy(), out[0][:n].cpu().numpy())
    np.testing.assert_array_equal(an[1][:n].cpu().numpy(), an[0][:n].cpu().numpy())
    det.set_dynamic_batch(None)  # back to static: the whole batch again
    det.predict(dev, B, size, size, 1, out[1], cf[1], an[1], conf=0.1)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out[1].cpu().numpy(), out[0].cpu().numpy())
    det.close()
