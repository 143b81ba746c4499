"""GPU tests at BASELINE.json's configurations 3, 4 and 5 (per-GPU shapes), and the accuracy statement of the
fp16 mode against the fp32 CPU restatement ("fp16 mode reports IoU distribution and
index-match rate"; the reference computes in fp32, yolo/yolo_train_config.yaml:51 `half: False`).

  config 3  full sim loop, YOLOv8s 640x640, 64 frames per step           -> test_fp16_accuracy_*, test_c4_*
  config 4  50 k frames sharded over 8 ranks (one rank's share here)       -> test_c4_one_rank_share_*
  config 5  1280x1280 fp16, 256 frames per step                            -> test_c5_*
"""
import os

import numpy as np
import pytest
import torch

from oracle import resmlp_oracle
from oracle import yolo_oracle as yo
from wtracker_amd import frames as fr
from wtracker_amd import hip, metrics, resmlp
from wtracker_amd import yolo_spec as ys
from wtracker_amd.pipeline import TrackPipeline

pytestmark = pytest.mark.gpu

# ---- the floors this build holds in fp16 mode at 640x640 (measured: 248 / 256 = 0.969 index match, every mismatch with an
# oracle top-1 / top-2 logit gap below 0.02; matched-frame IoU min 0.9974) -----------------------------------------------
F16_INDEX_MATCH_MIN = 0.96      # measured 250 / 256 = 0.977 (0.973 on a second set): one and a half sigma of head-room, a real regression fails
F16_IOU_MATCHED_MIN = 0.99      # every matched frame
F16_IOU_MATCHED_P01 = 0.995
F16_CONF_ATOL = 0.02


_ORACLE_CACHE: dict = {}


def _s_models(size, dtype, max_batch, seed=0):
    w = ys.synthetic_weights("s", 1, seed=seed)
    depth, width, maxch = ys.SCALES["s"]
    det = hip.HipYolo(w, (size, size), max_batch, dtype=dtype, nc=1, width=width, depth=depth, max_channels=maxch)
    return w, yo.YoloOracle(w, ys.model_dims(width, depth, maxch, 1)), det


def _oracle_run(oracle, frames, size, conf, chunk=16):
    xs, cs, an, gaps = [], [], [], []
    with torch.no_grad():
        for i in range(0, len(frames), chunk):
            x, hw = yo.preprocess(list(frames[i : i + chunk]), size)
            box, cls = oracle.forward(x)
            a, b, c = yo.postprocess(box, cls, tuple(x.shape[2:]), hw, conf=conf)
            xs.append(np.asarray(a, dtype=np.float64)), cs.append(b), an.append(c)
            top2 = torch.topk(cls.max(2).values, 2, dim=1).values
            gaps.append((top2[:, 0] - top2[:, 1]).numpy())
    return np.concatenate(xs), np.concatenate(cs), np.concatenate(an), np.concatenate(gaps)


def test_fp16_accuracy_vs_fp32_oracle_at_640_b64(hip_lib, capsys):
    """256 distinct synthetic 640x640 frames (64 tracks x 4 frames), fp16 mode in batches of 64 against the fp32 oracle:
    survivor-index match rate and IoU distribution are REPORTED (printed as one JSON line) and held above stated floors.
    The fp32 mode on the same frames must match the oracle's survivor on every frame."""
    import json

    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    size, N, B = 640, 256, 64
    w, oracle, det16 = _s_models(size, "fp16", B)
    frames = fr.diverse_frames(N, size, seed=1000)
    xo, co, ao, gap = _oracle_run(oracle, frames, size, 0.1)
    res = [det16.predict_host(frames[i : i + B], conf=0.1) for i in range(0, N, B)]
    det16.close()
    xg, cg, ag = (np.concatenate([r[k] for r in res]) for k in range(3))
    rep = metrics.accuracy_report(xg, ag, xo, ao, cg, co)
    bad = np.nonzero(ag != ao)[0]
    rep["mismatch_oracle_logit_gap_max"] = float(gap[bad].max()) if len(bad) else 0.0
    # NaN-row behaviour at the hardest threshold: the median of the oracle's best scores splits the frames in half
    t = float(np.median(co))
    rep["nan_row_agreement_at_median_conf"] = float(((cg > t) == (co > t)).mean())
    with capsys.disabled():
        print("\nfp16_accuracy_640_b64 " + json.dumps(rep))
    assert rep["checker_detections"] > N // 2
    assert rep["index_match_rate"] >= F16_INDEX_MATCH_MIN
    assert rep["iou_matched"]["min"] >= F16_IOU_MATCHED_MIN and rep["iou_matched"]["p01"] >= F16_IOU_MATCHED_P01
    assert rep["conf_abs_err_max"] <= F16_CONF_ATOL
    assert rep["nan_row_agreement"] >= 0.98 and rep["nan_row_agreement_at_median_conf"] >= 0.95
    # a differing survivor only ever happens where the oracle itself is nearly tied
    assert rep["mismatch_oracle_logit_gap_max"] < 0.03  # measured 0.019 at most (1 792 frames)
    # reference precision: the fp32 mode and the split-fp16 mode ("f16x3") are index-exact on all 256 frames
    for exact in ("fp32", "f16x3"):
        _, _, det32 = _s_models(size, exact, B)
        res = [det32.predict_host(frames[i : i + B], conf=0.1) for i in range(0, N, B)]
        det32.close()
        x32, c32, a32 = (np.concatenate([r[k] for r in res]) for k in range(3))
        np.testing.assert_array_equal(a32, ao)
        ok = ao >= 0
        np.testing.assert_allclose(x32[ok], xo[ok], rtol=0, atol=2e-2)
        np.testing.assert_allclose(c32, co, rtol=0, atol=1e-4)


def test_fp16_with_margin_recheck_restores_fp32_survivors(hip_lib, tmp_path):
    """YoloConfig(dtype="fp16", recheck_margin=0.08): the detector reports every frame's decision margin (best vs second-best anchor
    logit / distance to the conf threshold); frames below the margin are re-run by a full-precision handle (f16x3 at scale s).  On the 256-frame accuracy set the
    fp32 restatement's survivor must come back on EVERY frame, with a minority of the frames re-run; the margins themselves are
    checked against the oracle's top-2 logit gaps."""
    from wtracker_amd.controllers import HipYoloController, YoloConfig
    from wtracker_amd.sim import ExperimentConfig, TimingConfig

    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    size, N, B = 640, 256, 64
    w = ys.synthetic_weights("s", 1, seed=0)
    path = str(tmp_path / "s.wtk")
    ys.save_weights(path, w, "s", 1)
    depth, width, maxch = ys.SCALES["s"]
    oracle = yo.YoloOracle(w, ys.model_dims(width, depth, maxch, 1))
    frames = fr.diverse_frames(N, size, seed=1000)
    xo, co, ao, gap = _oracle_run(oracle, frames, size, 0.1)
    ec = ExperimentConfig("synthetic", N, 60, (size, size), 32, (size // 2, size // 2))
    tc = TimingConfig(ec, 100, 40, 50, (4, 4), (0.5, 0.5))
    plain = HipYoloController(tc, YoloConfig(model_path=path, pred_kwargs={"imgsz": size, "conf": 0.1}, dtype="fp16", max_batch=B))
    guarded = HipYoloController(tc, YoloConfig(model_path=path, pred_kwargs={"imgsz": size, "conf": 0.1}, dtype="fp16", max_batch=B, recheck_margin=0.08))
    rows_p, rows_g, rechecked, margins = [], [], 0, []
    for i in range(0, N, B):
        rows_p.append(np.asarray(plain.predict(list(frames[i : i + B])), dtype=np.float64))
        margins.append(plain._model.detector((size, size), B).last_margins(B))
        rows_g.append(np.asarray(guarded.predict(list(frames[i : i + B])), dtype=np.float64))
        rechecked += guarded.last_rechecked
    rows_p, rows_g, margins = np.concatenate(rows_p), np.concatenate(rows_g), np.concatenate(margins)
    # margins track the oracle's own: min(top-1 / top-2 logit gap, distance of the best logit from logit(conf)), within the fp16 logit noise
    best_logit = np.log(co.astype(np.float64) / (1.0 - co.astype(np.float64)))
    want = np.minimum(gap, np.abs(best_logit - np.log(0.1 / 0.9)))
    assert np.abs(margins - want).max() < 0.1 and np.median(np.abs(margins - want)) < 0.01 and (margins >= 0).all()
    ok = ao >= 0
    np.testing.assert_allclose(rows_g[ok], xo[ok], rtol=0, atol=1.0)  # the fp32 survivor's box (fp16 rows: IoU >= 0.997 -> well inside 1 px)
    close = np.abs(rows_g[ok] - xo[ok]).max(axis=1) < 1.0
    assert close.all()
    n_plain_off = int((np.abs(rows_p[ok] - xo[ok]).max(axis=1) >= 1.0).sum())
    assert 0 < rechecked < 0.4 * N, rechecked
    assert guarded.yolo_config.recheck_mode() == "f16x3"  # scale s: the second look runs on the fp16 matrix pipe with split operands
    print(f"\nrecheck: {rechecked} of {N} frames re-run in f16x3; plain fp16 rows off by >= 1 px: {n_plain_off}, guarded: 0")


@pytest.mark.parametrize("dtype,B", [("fp16", 2), ("fp32", 1)])
def test_c5_full_size_1280_matches_oracle(hip_lib, dtype, B):
    """BASELINE config 5 frame shape (1280x1280, A = 33 600 anchors) against the restatement: head logits within the
    mode's tolerance, survivor index equal (fp32) / explained by the measured logit error (fp16, no escape)."""
    from test_gpu_yolo import F32_BOX_ATOL, F32_LOGIT_ATOL, _assert_fp16_survivors_explained

    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    size = 1280
    w, oracle, det = _s_models(size, dtype, 2)
    assert det.anchors == 33600 and abs(det.macs_per_frame - 4 * 14.2158336e9) < 1e4
    frames = fr.diverse_frames(B, size, seed=500, per_seed=1)
    with torch.no_grad():
        x, hw = yo.preprocess(list(frames), size)
        box_o, cls_o = oracle.forward(x)
    xywh, conf, anchor = det.predict_host(frames, conf=0.1)
    box_g, cls_g = det.debug_head(B)
    if dtype == "fp32":
        assert np.abs(cls_g - cls_o.numpy()).max() < F32_LOGIT_ATOL and np.abs(box_g - box_o.numpy()).max() < F32_LOGIT_ATOL
        xywh_o, _, anchor_o = yo.postprocess(box_o, cls_o, (size, size), hw, conf=0.1)
        np.testing.assert_array_equal(anchor, anchor_o)
        np.testing.assert_allclose(xywh, xywh_o, rtol=0, atol=F32_BOX_ATOL)
    else:
        _assert_fp16_survivors_explained(anchor, xywh, box_g, cls_g, box_o, cls_o, (size, size), hw, 0.1)
    det.close()


def test_c5_batch256_1280_properties(hip_lib):
    """BASELINE config 5 per-GPU shape (256 frames of 1280x1280, fp16): determinism, batch invariance (sub-batches of 32 and
    a permuted batch give bit-identical rows), boxes inside the frame, and agreement of the first rows with a B=2 handle."""
    size, B = 1280, 256
    w, _, det = _s_models(size, "fp16", B)
    pool = fr.diverse_frames(32, size, seed=700, per_seed=2)
    rng = np.random.default_rng(0)
    frames = pool[rng.integers(0, len(pool), size=B)]
    frames[1::2] = frames[1::2, ::-1, :]  # flipped copies: 64 distinct images without drawing 256 frames on the host
    a = det.predict_host(frames, conf=0.1)
    b = det.predict_host(frames, conf=0.1)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    sub = [det.predict_host(frames[i : i + 32], conf=0.1) for i in range(0, B, 32)]
    np.testing.assert_array_equal(np.concatenate([s[0] for s in sub]), a[0])
    np.testing.assert_array_equal(np.concatenate([s[2] for s in sub]), a[2])
    perm = rng.permutation(B)
    p = det.predict_host(frames[perm], conf=0.1)
    np.testing.assert_array_equal(p[0], a[0][perm])
    np.testing.assert_array_equal(p[2], a[2][perm])
    ok = a[2] >= 0
    assert ok.sum() > 0 and (a[0][ok, 2:] > 0).all()
    assert (a[0][ok, :2] >= 0).all() and (a[0][ok, 0] + a[0][ok, 2] <= size + 1e-3).all() and (a[0][ok, 1] + a[0][ok, 3] <= size + 1e-3).all()
    det.close()
    _, _, small = _s_models(size, "fp16", 2)
    s2 = small.predict_host(frames[:2], conf=0.1)
    np.testing.assert_array_equal(s2[0], a[0][:2])
    np.testing.assert_array_equal(s2[2], a[2][:2])
    small.close()


def test_c4_one_rank_share_two_lanes_equal_one_lane_and_oracle_moves(hip_lib, golden_dir):
    """BASELINE config 4, one rank's share at full model size: TrackPipeline over 6 144 frames of 640x640 (YOLOv8s fp16,
    96 steps of 64 frames = what one of 8 ranks detects of 50 k frames), two lanes in flight vs one lane: the whole track,
    every validity flag and every ResMLP move bit-identical; every move equal to resmlp_oracle on that track."""
    size, B, steps = 640, 64, 96
    path = os.path.join(golden_dir, "resmlp_100ms.npz")
    folded = resmlp.load_npz(path)
    w = ys.synthetic_weights("s", 1, seed=0)
    depth, width, maxch = ys.SCALES["s"]
    pool_np = fr.diverse_frames(512, size, seed=300, per_seed=8)
    pool = torch.from_numpy(pool_np).cuda()
    order = np.random.default_rng(5).permutation(steps * B) % len(pool_np)
    order_dev = torch.from_numpy(order).cuda()

    def run(lanes):
        dets = [hip.HipYolo(w, (size, size), B, dtype="fp16", nc=1, width=width, depth=depth, max_channels=maxch) for _ in range(lanes)]
        mlp = hip.HipMLP(folded.layers, folded.n_blocks, folded.layers_per_block)
        pipe = TrackPipeline(dets, mlp, folded, B, steps * B, imaging_frame_num=6, pred_frame_num=3, cycle_frame_num=9, conf=0.1)
        bufs = [torch.empty((B, size, size), dtype=torch.uint8, device="cuda") for _ in range(2 * lanes)]
        for s in range(steps):
            buf = bufs[s % len(bufs)]
            torch.index_select(pool, 0, order_dev[s * B : (s + 1) * B], out=buf)
            pipe.step(s, buf)
            if s % len(bufs) == len(bufs) - 1:
                pipe.synchronize()  # the gather buffers are about to be overwritten
        pipe.synchronize()
        torch.cuda.synchronize()
        out = pipe.track.cpu().numpy(), pipe.moves.cpu().numpy(), pipe.valid.cpu().numpy(), pipe.plan
        for d in dets:
            d.close()
        return out

    t1, m1, v1, plan = run(1)
    t2, m2, v2, _ = run(2)
    np.testing.assert_array_equal(t1, t2)
    np.testing.assert_array_equal(v1, v2)
    np.testing.assert_array_equal(m1, m2)
    # one prediction per 9-frame cycle whose provide_movement_vector frame (anchor + pred_frame_num) exists: 3, 12, ..., 6132
    assert len(plan.anchors) == (steps * B - 6 - 1) // 9 + 1 == 682 and plan.steps == steps
    # the track itself: a frame's row only depends on the frame (same pool image -> same row)
    first = {}
    for f, src in enumerate(order):
        if src in first:
            np.testing.assert_array_equal(t1[f], t1[first[src]])
        else:
            first[src] = f
    # every ResMLP move against the oracle arithmetic on the same fp32 track
    st = resmlp_oracle.load_state(path)
    inf = np.asarray(folded.input_frames)
    idx = plan.anchors[:, None] + inf[None, :]
    inside = (idx >= 0).all(axis=1)
    rows = t1[np.clip(idx, 0, len(t1) - 1)]                      # [n,7,4]
    ok = inside & np.isfinite(rows).all(axis=(1, 2))
    np.testing.assert_array_equal(v1.astype(bool), ok)
    x = rows[ok].copy()
    x[:, :, 0] -= x[:, :1, 0]
    x[:, :, 1] -= x[:, :1, 1]
    expect = resmlp_oracle.forward(st, x.reshape(len(x), -1).astype(np.float32))
    assert ok.sum() > 600
    np.testing.assert_allclose(m1[ok], expect, rtol=1e-4, atol=5e-4)


@pytest.mark.parametrize("margin,K", [(0.08, 24), (0.04, 16)])
def test_hybrid_detector_fp16_speed_with_full_precision_decisions(hip_lib, margin, K):
    """wtracker_amd.hybrid.HybridDetector on the 256-frame accuracy set: fp16 on every frame, the K weakest decisions of each 64-frame
    batch again through the split-fp16 ("f16x3") handle, all on the device (wtk_recheck_select / wtk_recheck_merge, no host round
    trip).  Every survivor index must equal the fp32 restatement's; rows the second look did not replace are the fp16 rows bit for
    bit; the views entry point gives the same rows as the frames entry point."""
    from wtracker_amd.hybrid import HybridDetector

    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    size, N, B = 640, 256, 64
    w, oracle, fast = _s_models(size, "fp16", B)
    _, _, exact = _s_models(size, "f16x3", K)
    _, _, plain = _s_models(size, "fp16", B)
    frames = fr.diverse_frames(N, size, seed=1000)
    if "hybrid" not in _ORACLE_CACHE:
        _ORACLE_CACHE["hybrid"] = _oracle_run(oracle, frames, size, 0.1)
    xo, co, ao, gap = _ORACLE_CACHE["hybrid"]
    hyb = HybridDetector(fast, exact, margin=margin, k=K)
    dev = torch.from_numpy(frames).cuda()
    out = torch.empty((N, 4), dtype=torch.float32, device="cuda")
    cf = torch.empty((N,), dtype=torch.float32, device="cuda")
    an = torch.empty((N,), dtype=torch.int32, device="cuda")
    out_p, an_p = torch.empty_like(out), torch.empty_like(an)
    margins = []
    for i in range(0, N, B):
        hyb.predict(dev[i : i + B], B, size, size, 1, out[i : i + B], cf[i : i + B], an[i : i + B], conf=0.1)
        plain.predict(dev[i : i + B], B, size, size, 1, out_p[i : i + B], None, an_p[i : i + B], conf=0.1)
        torch.cuda.synchronize()
        margins.append(plain.last_margins(B))
    margins = np.concatenate(margins)
    xg, ag, xp, ap = out.cpu().numpy(), an.cpu().numpy(), out_p.cpu().numpy(), an_p.cpu().numpy()
    replaced = int(hyb.replaced.item())
    np.testing.assert_array_equal(ag, ao)  # the reference precision's survivor on every frame
    ok = ao >= 0
    assert np.abs(xg[ok] - xo[ok]).max() < 1.0
    strong = margins >= margin
    np.testing.assert_array_equal(xg[strong], xp[strong])  # untouched rows are the fp16 rows
    expected = sum(min(K, int((margins[i : i + B] < margin).sum())) for i in range(0, N, B))
    assert replaced == expected and 0 < replaced < 0.4 * N, (replaced, expected)
    n_plain_bad = int((ap != ao).sum())
    print(f"\nhybrid (margin {margin}): {replaced} of {N} rows replaced by the f16x3 look (K = {K} per batch of {B}); plain fp16 index mismatches {n_plain_bad}, hybrid 0")
    # views entry point: whole-frame views in shuffled order
    perm = np.random.default_rng(0).permutation(B).astype(np.int32)
    idx = torch.from_numpy(perm).cuda()
    pos = torch.tensor([[size // 2, size // 2]] * B, dtype=torch.int32).cuda()
    out_v, an_v = torch.empty((B, 4), dtype=torch.float32, device="cuda"), torch.empty((B,), dtype=torch.int32, device="cuda")
    hyb.predict_views(dev[:B], B, size, size, 1, idx, pos, B, size, size, out_v, None, an_v, conf=0.1)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(an_v.cpu().numpy(), ag[:B][perm])
    np.testing.assert_array_equal(out_v.cpu().numpy(), xg[:B][perm])
    hyb.close()
    plain.close()
