"""GPU: the hybrid detector (fp16 + device-side full-precision second look, wtracker_amd/hybrid.py) validated OUT OF SAMPLE, and its
ceiling accounted for.

The reference looks at every frame in fp32 and keeps the arg-max anchor (yolo/yolo_train_config.yaml:51, yolo_controller.py:72-90);
the hybrid's claim is the same survivor on every frame.  The threshold 0.04 came from tools/margin_study.py (frame seeds 1000-9127,
weight seed 0); here: 2 048 frames of seeds 20000-20511 for EACH of the weight seeds 1, 2, 3 — none of them seen by that study."""
import numpy as np
import pytest
import torch

from wtracker_amd import frames as fr

pytestmark = pytest.mark.gpu

MARGIN = 0.04
_FRAMES: dict = {}


def _frames(n, seed=20000):
    if (n, seed) not in _FRAMES:
        _FRAMES[(n, seed)] = fr.diverse_frames(n, 640, seed=seed)
    return _FRAMES[(n, seed)]


@pytest.mark.parametrize("weight_seed", [1, 2, 3])
def test_hybrid_out_of_sample_2048_frames_per_weight_seed(hip_lib, weight_seed, capsys):
    import json
    import sys, os

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from hybrid_validation import run_weight_seed

    frames = _frames(2048)
    rep = run_weight_seed(weight_seed, frames, 640, 64, MARGIN, 0.1, oracle_frames=128)
    with capsys.disabled():
        print(f"\nhybrid_validation weight_seed={weight_seed} " + json.dumps(rep))
    assert rep["frames"] == 2048
    # the synthetic weights of this seed exercise both outcomes (detection and NaN row) — the stored gains transfer
    assert 0.2 * 2048 < rep["detections_f16x3"] <= 2048
    # THE claim: every frame's survivor is the full-precision handle's, no weak row was cut off
    assert rep["hybrid_equals_f16x3_index"] and rep["hybrid_index_mismatches_vs_f16x3"] == 0
    assert rep["hybrid_overflow_rows"] == 0 and rep["ceiling_per_batch"] == 64
    assert rep["hybrid_strong_rows_are_fp16_rows"] and rep["hybrid_weak_rows_are_f16x3_rows"]
    # why it holds: fp16 alone does differ on some frames, and every such frame sits well below the threshold
    assert rep["fp16_mismatch_margin_max"] < 0.75 * MARGIN, rep["fp16_mismatch_margins_sorted_desc"]
    assert rep["hybrid_rows_replaced"] == rep["frames_below_threshold"] and rep["share_below_threshold"] < 0.4
    # against the fp32 CPU restatement on the first 128 frames: the exact modes pick its survivor on every frame
    for mode in ("f16x3", "hybrid"):
        r = rep[f"vs_cpu_restatement_{mode}"]
        assert r["index_match_rate"] == 1.0, (mode, r)
        assert r["iou_matched"] is None or r["iou_matched"]["min"] > 0.999


def test_hybrid_ceiling_smaller_than_the_weak_rows_reports_overflow(hip_lib):
    """K < number of weak rows: the K weakest rows get the full-precision result, the rest keep their fp16 rows AND the
    overflow counter says how many (wtk_recheck_select_counted) — the shortfall is never silent."""
    from wtracker_amd import hip
    from wtracker_amd import yolo_spec as ys
    from wtracker_amd.hybrid import HybridDetector

    size, B, K, wide = 640, 64, 8, 0.5  # a wide margin makes most rows "weak"
    w = ys.synthetic_weights("s", 1, seed=0)
    depth, width, maxch = ys.SCALES["s"]
    mk = lambda dt, mb: hip.HipYolo(w, (size, size), mb, dtype=dt, nc=1, width=width, depth=depth, max_channels=maxch)
    plain, exact = mk("fp16", B), mk("f16x3", B)
    hyb = HybridDetector(mk("fp16", B), mk("f16x3", K), margin=wide, k=K)
    frames = torch.from_numpy(_frames(2048)[: 2 * B]).cuda()
    expect_overflow, weak_first = 0, None
    for i in range(0, 2 * B, B):
        o = {n: (torch.empty((B, 4), device="cuda"), torch.empty((B,), device="cuda"), torch.empty((B,), dtype=torch.int32, device="cuda")) for n in ("p", "e", "h")}
        for n, det in (("p", plain), ("e", exact), ("h", hyb)):
            det.predict(frames[i : i + B], B, size, size, 1, *o[n], conf=0.1)
        torch.cuda.synchronize()
        m = plain.last_margins(B)
        n_weak = int((m < wide).sum())
        weak_first = n_weak if weak_first is None else weak_first
        assert n_weak > K
        expect_overflow += n_weak - K
        order = np.lexsort((np.arange(B), m))  # margin ascending, ties by row
        took = np.zeros(B, dtype=bool)
        took[order[:K]] = True
        xh, xe, xp = (o[n][0].cpu().numpy() for n in ("h", "e", "p"))
        np.testing.assert_array_equal(xh[took], xe[took])    # the K weakest rows: the full-precision rows
        np.testing.assert_array_equal(xh[~took], xp[~took])  # everything else: the fp16 rows, second look or not
    assert hyb.overflow_count() == expect_overflow > 0
    assert int(hyb.replaced.item()) == 2 * K
    # the default ceiling is the batch: the same data cannot overflow
    full = HybridDetector(mk("fp16", B), mk("f16x3", B), margin=wide)
    assert full.k == B
    out = (torch.empty((B, 4), device="cuda"), torch.empty((B,), device="cuda"), torch.empty((B,), dtype=torch.int32, device="cuda"))
    full.predict(frames[:B], B, size, size, 1, *out, conf=0.1)
    torch.cuda.synchronize()
    assert full.overflow_count() == 0 and int(full.replaced.item()) == weak_first
    for d in (plain, exact):
        d.close()
    hyb.close()
    full.close()
