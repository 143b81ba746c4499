"""GPU parity: ResMLP kernel + HipMLPController vs the reference's golden vectors and the oracle.
Everything goes through the C ABI (ctypes -> libwtk_hip.so)."""
import json
import os

import numpy as np
import pytest

from oracle import resmlp_oracle
from wtracker_amd import hip, resmlp
from wtracker_amd.controllers import HipMLPController
from wtracker_amd.sim import ExperimentConfig, Simulator, TimingConfig, TrackLogger

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
