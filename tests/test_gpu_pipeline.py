"""GPU: the batched open-loop pipeline (detect -> track -> ResMLP per cycle), one lane vs three lanes in
flight, and its ResMLP outputs vs the oracle arithmetic on the same track."""
import os

import numpy as np
import pytest
import torch

from oracle import resmlp_oracle
from wtracker_amd import frames as fr
from wtracker_amd import hip, resmlp
from wtracker_amd import yolo_spec as ys
from wtracker_amd.pipeline import TrackPipeline

pytestmark = pytest.mark.gpu


def _run(lanes, frames, B, steps, folded, weights):
    dets = [hip.HipYolo(weights, (128, 128), B, dtype="fp16", nc=1, width=0.25, depth=0.33, max_channels=1024) for _ in range(lanes)]
    mlp = hip.HipMLP(folded.layers, folded.n_blocks, folded.layers_per_block)
    pipe = TrackPipeline(dets, mlp, folded, B, steps * B, imaging_frame_num=6, pred_frame_num=3, cycle_frame_num=9, conf=0.1)
    for s in range(steps):
        pipe.step(s, frames[s * B : (s + 1) * B])
    pipe.synchronize()
    torch.cuda.synchronize()
    global _last_pipe
    _last_pipe = pipe
    return pipe.track.cpu().numpy(), pipe.moves.cpu().numpy(), pipe.valid.cpu().numpy(), pipe.plan


def test_three_lanes_equal_one_lane_and_oracle_mlp(hip_lib, golden_dir):
    B, steps = 32, 6
    path = os.path.join(golden_dir, "resmlp_100ms.npz")
    folded = resmlp.load_npz(path)
    weights = ys.synthetic_weights("n", 1, seed=0)
    f_np, _ = fr.synthetic_frames(B * steps, 128, seed=4)
    frames = torch.from_numpy(f_np).cuda()
    t1, m1, v1, plan = _run(1, frames, B, steps, folded, weights)
    t3, m3, v3, _ = _run(3, frames, B, steps, folded, weights)
    np.testing.assert_array_equal(t1, t3)
    np.testing.assert_array_equal(v1, v3)
    np.testing.assert_array_equal(m1, m3)
    assert np.isfinite(t1).all() or np.isnan(t1).any()
    # ResMLP outputs == oracle arithmetic on the same (fp32) track
    st = resmlp_oracle.load_state(path)
    n_valid = 0
    for i, a in enumerate(plan.anchors):
        idx = a + np.asarray(folded.input_frames)
        ok = (idx >= 0).all() and np.isfinite(t1[np.clip(idx, 0, len(t1) - 1)]).all()
        assert bool(v1[i]) == bool(ok)
        if ok:
            b = t1[idx].copy()
            b[:, 0] -= b[0, 0]
            b[:, 1] -= b[0, 1]
            np.testing.assert_allclose(m1[i], resmlp_oracle.forward(st, b.reshape(1, -1))[0], rtol=1e-4, atol=5e-4)
            n_valid += 1
    assert n_valid >= 10 and len(plan.anchors) == (B * steps - 6) // 9 + 1


def test_pipeline_baseline_targets_match_numpy_on_the_device_track(hip_lib, golden_dir):
    """after an open-loop run all three predictors' outputs exist on the device without a host pass — ResMLP moves
    (above) plus OptimalController / PolyfitController targets for every cycle, here against numpy on the same fp32 track."""
    from numpy.polynomial import polynomial as poly

    B, steps = 32, 4
    folded = resmlp.load_npz(os.path.join(golden_dir, "resmlp_100ms.npz"))
    f_np, _ = fr.synthetic_frames(B * steps, 128, seed=6)
    t, _, _, plan = _run(2, torch.from_numpy(f_np).cuda(), B, steps, folded, ys.synthetic_weights("n", 1, seed=0))
    times, weights = [-9, -6, -3, 0, 2, 4], [1, 1, 2, 3, 4, 5.0]
    out = _last_pipe.baseline_targets(sample_times=times, weights=weights, degree=2)
    torch.cuda.synchronize()
    cen = np.stack([t[:, 0].astype(np.float64) + t[:, 2].astype(np.float64) / 2, t[:, 1].astype(np.float64) + t[:, 3].astype(np.float64) / 2], axis=1)
    opt, opt_ok = (x.cpu().numpy() for x in out["optimal"])
    pf, pf_ok = (x.cpu().numpy() for x in out["polyfit"])
    assert len(opt) == len(plan.anchors) and opt_ok.sum() >= len(opt) - 2
    for c in range(len(opt)):
        w = cen[(c + 1) * 9 : (c + 1) * 9 + 6]
        w = w[np.isfinite(w).all(axis=1)]
        assert bool(opt_ok[c]) == (len(w) > 0)
        if len(w):
            np.testing.assert_array_equal(opt[c], np.median(w, axis=0))
        f = c * 9 + np.asarray(times)
        ok = (f >= 0) & (f < len(t))
        ok[ok] &= np.isfinite(cen[f[ok]]).all(axis=1)
        assert bool(pf_ok[c]) == bool(ok.any())
        if ok.sum() >= 3:
            coef = poly.polyfit(np.asarray(times)[ok], cen[f[ok]], deg=2, w=np.asarray(weights)[ok])
            np.testing.assert_allclose(pf[c], poly.polyval(9 + 6 // 2, coef), rtol=0, atol=1e-6)


def test_c_abi_communicator_single_rank_allgather(hip_lib):
    """wtk_comm_* / wtk_allgather_tracks (the C ABI's own RCCL communicator) in the degenerate one-rank world a
    one-GPU box allows: rendezvous token, communicator, an all-gather that must reproduce the local slice bit for bit
    (NaN rows included), on a side stream.  RCCL refuses two ranks on one device, so N > 1 is covered by the gloo tests."""
    uid = hip.comm_unique_id()
    assert len(uid) == hip.COMM_ID_BYTES and any(uid)
    comm = hip.WtkComm(0, 0, 1, uid)
    local = torch.rand((64, 4), device="cuda")
    local[5] = float("nan")
    out = torch.zeros((64, 4), device="cuda")
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        comm.allgather_tracks(local, 64, out, stream=st.cuda_stream)
    st.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), local.cpu().numpy())
    with pytest.raises(hip.WtkError, match="rank"):
        hip.WtkComm(0, 2, 2, uid)
    comm.close()
