"""GPU parity: ResMLP kernel + HipMLPController vs the reference's golden vectors and the oracle.
Everything goes through the C ABI (ctypes -> libwtk_hip.so)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import resmlp_oracle
from wtracker_amd import hip, resmlp
from wtracker_amd.controllers import HipMLPController
from wtracker_amd.sim import ExperimentConfig, TimingConfig, TrackLogger
from harness.sim_harness import Simulator

pytestmark = pytest.mark.gpu
ATOL, RTOL = 2e-4, 1e-5
EXP0 = dict(name="exp0", num_frames=200, frames_per_sec=60, orig_resolution=(1600, 1400), px_per_mm=90, init_position=(1300, 1200))


def _mlp(golden_dir, tag):
    m = resmlp.load_npz(os.path.join(golden_dir, f"resmlp_{tag}.npz"))
    return m, hip.HipMLP(m.layers, m.n_blocks, m.layers_per_block)


@pytest.mark.parametrize("tag", ["100ms", "200ms"])
def test_forward_matches_reference_golden(hip_lib, golden_dir, tag):
    z = np.load(os.path.join(golden_dir, f"resmlp_{tag}.npz"))
    _, g = _mlp(golden_dir, tag)
    y = g.forward_host(z["x"])
    np.testing.assert_allclose(y, z["y_batch"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(g.forward_host(np.zeros((1, 28), np.float32)), z["y_zero"], rtol=RTOL, atol=ATOL)
    # ragged batches: 1, 15, 16, 17 samples give the same rows (one wave = 16 samples)
    for n in (1, 15, 16, 17, 33):
        np.testing.assert_array_equal(g.forward_host(z["x"][:n]), y[:n])
    assert g.forward_host(np.zeros((0, 28), np.float32)).shape == (0, 2)


@pytest.mark.parametrize("tag", ["100ms", "200ms"])
def test_forward_matches_oracle_on_large_batch(hip_lib, golden_dir, tag):
    st = resmlp_oracle.load_state(os.path.join(golden_dir, f"resmlp_{tag}.npz"))
    _, g = _mlp(golden_dir, tag)
    rng = np.random.default_rng(5)
    x = rng.normal(0, 6, size=(5556, 28)).astype(np.float32)  # BASELINE C4: 50 000 / 9 cycles
    np.testing.assert_allclose(g.forward_host(x), resmlp_oracle.forward(st, x), rtol=1e-4, atol=5e-4)


@pytest.mark.parametrize("tag,timing,name", [("100ms", (100, 40, 50), "sim_mlp_bboxes.csv"), ("200ms", (200, 40, 50), "sim_mlp200_bboxes.csv")])
def test_controller_sim_loop_matches_reference_log(hip_lib, golden_dir, tag, timing, name):
    """BASELINE config 1: simulate.ipynb CsvController+ResMLP loop on 200 pre-detected frames; integer
    (dx, dy) per cycle and every logged row equal to what the real reference produced."""
    import csv

    ec = ExperimentConfig(**EXP0)
    tc = TimingConfig(ec, *timing, (4, 4), (0.32, 0.32))
    m = resmlp.load_npz(os.path.join(golden_dir, f"resmlp_{tag}.npz"))
    ctrl = HipMLPController(tc, os.path.join(golden_dir, "sim_init_bboxes.csv"), m, max_speed=0.9)
    moves = []
    inner = ctrl.provide_movement_vector

    def wrapped(sim):
        dx, dy = inner(sim)
        moves.append([int(sim.frame_number), int(dx), int(dy)])
        return dx, dy

    ctrl.provide_movement_vector = wrapped
    log = TrackLogger(ctrl)
    Simulator(tc, ec, log).run()
    assert moves == json.load(open(os.path.join(golden_dir, "sim_moves.json")))[name]
    golden = list(csv.DictReader(open(os.path.join(golden_dir, name), newline="")))
    assert len(log.rows) == len(golden)
    for r, g in zip(log.rows, golden):
        for k in ("frame", "cycle", "plt_x", "plt_y", "cam_x", "cam_y", "mic_x", "mic_y", "wrm_x", "wrm_y", "wrm_w", "wrm_h"):
            assert float(r[k]) == float(g[k]), (k, g["frame"])


def test_predict_track_gather_matches_host_controller_arithmetic(hip_lib, golden_dir):
    import torch

    tag = "100ms"
    st = resmlp_oracle.load_state(os.path.join(golden_dir, f"resmlp_{tag}.npz"))
    m, g = _mlp(golden_dir, tag)
    rng = np.random.default_rng(3)
    n = 400
    track = np.cumsum(rng.normal(0, 0.6, size=(n, 4)), axis=0).astype(np.float32)
    track[:, 2:] = 14 + rng.normal(0, 0.5, size=(n, 2))
    track[:, :2] += 300
    track[123] = np.nan  # a missed detection poisons every sample that gathers it
    anchors = np.arange(-5, n + 5, 9, dtype=np.int32)
    t_dev = torch.from_numpy(track).cuda()
    a_dev = torch.from_numpy(anchors).cuda()
    pred = torch.empty((len(anchors), 2), dtype=torch.float32, device="cuda")
    valid = torch.empty((len(anchors),), dtype=torch.int32, device="cuda")
    g.predict_track(t_dev, n, a_dev, len(anchors), m.input_frames, pred, valid, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    pred, valid = pred.cpu().numpy(), valid.cpu().numpy()
    for s, t in enumerate(anchors):
        idx = t + np.asarray(m.input_frames)
        ok = (idx >= 0).all() and (idx < n).all() and np.isfinite(track[np.clip(idx, 0, n - 1)]).all()
        assert bool(valid[s]) == bool(ok)
        if ok:
            b = track[idx].astype(np.float32).copy()
            b[:, 0] -= b[0, 0]
            b[:, 1] -= b[0, 1]
            ref = resmlp_oracle.forward(st, b.reshape(1, -1))[0]
            np.testing.assert_allclose(pred[s], ref, rtol=1e-4, atol=5e-4)
        else:
            assert (pred[s] == 0).all()
    assert valid.sum() > 10 and (valid == 0).sum() > 3


def test_device_optimal_and_polyfit_controllers_match_reference_moves(hip_lib, golden_dir):
    """SURVEY.md §8 f4 on the device: every cycle's OptimalController median / PolyfitController weighted fit computed in one
    launch over the device-resident track (wtk_track_median_centers / wtk_track_polyfit); the closed-loop integer moves equal
    what the REAL reference's controllers returned (sim_moves.json, polyfit_cases.json incl. unsorted times + weights)."""
    import json

    from test_sim_golden import run
    from wtracker_amd.controllers import HipOptimalController, HipPolyfitController, PolyfitConfig

    init = os.path.join(golden_dir, "sim_init_bboxes.csv")
    gold = json.load(open(os.path.join(golden_dir, "sim_moves.json")))
    _, moves = run(lambda tc: HipOptimalController(tc, init))
    assert moves == gold["sim_optimal"] and any(m[1:] != [0, 0] for m in moves)
    cfg = PolyfitConfig(degree=2, sample_times=[-9, -6, -3, 0, 2, 4], weights=[1, 1, 2, 3, 4, 5])
    _, moves = run(lambda tc: HipPolyfitController(tc, cfg, init))
    assert moves == gold["sim_polyfit"]
    for name, c in json.load(open(os.path.join(golden_dir, "polyfit_cases.json"))).items():
        cfg = PolyfitConfig(**c["config"])
        _, moves = run(lambda tc: HipPolyfitController(tc, cfg, init))
        assert moves == c["moves"], name
    # the widest problems the device solver admits (degree 7 x 16 times, scaled-Vandermonde condition ~1e9): the one-sided Jacobi SVD
    # keeps numpy's rcond = len(t) * eps cut-off, so the reference's integer moves come back here too
    for name, c in json.load(open(os.path.join(golden_dir, "polyfit_highdeg.json"))).items():
        cfg = PolyfitConfig(**c["config"])
        _, moves = run(lambda tc: HipPolyfitController(tc, cfg, init))
        diff = [(a, b) for a, b in zip(moves, c["moves"]) if a != b]
        assert not diff, (name, diff[:4])


def test_device_polyfit_and_median_against_numpy_on_ragged_tracks(hip_lib):
    """Raw predictor outputs (before rounding) against numpy on a track with runs of NaN rows, so that cycles with fewer samples
    than coefficients (minimum-norm solution), a single sample, and no sample at all occur; float32 and float64 tracks."""
    from numpy.polynomial import polynomial as poly

    rng = np.random.default_rng(3)
    n, cyc, img = 400, 9, 6
    t = np.cumsum(rng.normal(0.5, 0.3, size=(n, 2)), axis=0) + [700.0, 500.0]
    track = np.concatenate([t, 14 + rng.normal(0, 0.5, size=(n, 2))], axis=1)
    track[rng.random(n) < 0.25] = np.nan
    track[100:140] = np.nan
    centers = np.stack([track[:, 0] + track[:, 2] / 2, track[:, 1] + track[:, 3] / 2], axis=1)
    n_cycles = n // cyc + 2
    times, weights, deg = np.array([-9, -6, -3, 0, 2, 4]), np.array([1, 1, 2, 3, 4, 5.0]), 3
    for dtype in (torch.float64, torch.float32):
        tr = torch.from_numpy(track).to(dtype).cuda()
        trn = tr.double().cpu().numpy()
        cen = np.stack([trn[:, 0] + trn[:, 2] / 2, trn[:, 1] + trn[:, 3] / 2], axis=1) if dtype == torch.float32 else centers
        cycles = torch.arange(n_cycles, dtype=torch.int32, device="cuda")
        pred = torch.zeros((n_cycles, 2), dtype=torch.float64, device="cuda")
        valid = torch.zeros((n_cycles,), dtype=torch.int32, device="cuda")
        hip.track_median_centers(tr, n, cycles, n_cycles, cyc, img, pred, valid)
        torch.cuda.synchronize()
        p, v = pred.cpu().numpy(), valid.cpu().numpy()
        seen = set()
        for c in range(n_cycles):
            w = cen[(c + 1) * cyc : (c + 1) * cyc + img]
            w = w[np.isfinite(w).all(axis=1)]
            assert bool(v[c]) == (len(w) > 0)
            seen.add(min(len(w), 2))
            if len(w):
                np.testing.assert_array_equal(p[c], np.median(w, axis=0))  # bit-exact: sort + mean of the middle pair
        assert seen == {0, 1, 2}
        hip.track_polyfit(tr, n, cycles, n_cycles, cyc, times, weights, deg, cyc + img // 2, pred, valid)
        torch.cuda.synchronize()
        p, v = pred.cpu().numpy(), valid.cpu().numpy()
        ranks = set()
        for c in range(n_cycles):
            f = c * cyc + times
            ok = (f >= 0) & (f < n)
            ok[ok] &= np.isfinite(cen[f[ok]]).all(axis=1)
            assert bool(v[c]) == bool(ok.any())
            if ok.any():
                import warnings

                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")  # RankWarning on under-determined cycles: numpy returns the minimum-norm fit
                    coef = poly.polyfit(times[ok], cen[f[ok]], deg=deg, w=weights[ok])
                want = poly.polyval(cyc + img // 2, coef)
                ranks.add(min(int(ok.sum()), deg + 1))
                np.testing.assert_allclose(p[c], want, rtol=0, atol=1e-6)
        assert ranks == {1, 2, 3, 4}  # under-determined cycles are exercised


def test_device_training_pairs_match_reference_dataset(hip_lib, golden_dir):
    """NumpyDataset.create_from_config on the device (wtk_track_training_pairs) == the REAL reference's X / y, bit for bit."""
    import pandas as pd

    from wtracker_amd.resmlp import make_training_pairs_device

    g = np.load(os.path.join(golden_dir, "dataset_100ms.npz"))
    m = np.load(os.path.join(golden_dir, "resmlp_100ms.npz"))
    boxes = pd.read_csv(os.path.join(golden_dir, "sim_init_bboxes.csv"))[["wrm_x", "wrm_y", "wrm_w", "wrm_h"]].to_numpy(dtype=np.float64)
    X, y = make_training_pairs_device(torch.from_numpy(boxes).cuda(), m["input_frames"].tolist(), m["pred_frames"].tolist())
    assert X.dtype == torch.float32 and tuple(X.shape) == g["X"].shape and tuple(y.shape) == g["y"].shape
    assert np.array_equal(X.cpu().numpy(), g["X"]) and np.array_equal(y.cpu().numpy(), g["y"])
    X0, y0 = make_training_pairs_device(torch.from_numpy(boxes).cuda(), [-300, 0], [400])
    assert tuple(X0.shape) == (0, 8) and tuple(y0.shape) == (0, 2)
