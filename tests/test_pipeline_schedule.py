"""CPU: the SCHEDULE of wtracker_amd.pipeline.TrackPipeline — which step's exchange and ResMLP run when — with stand-in detector and
predictor objects (no GPU, no libwtk_hip.so call).  What is under test is host logic only:

  * a detector that holds rows back (HybridDetector(defer = D): the rows of a call are final only after the lane's next flush) must not
    let the ResMLP of a step run before the rows of that step AND of the step it looks back into are final, whatever the lane count;
  * the all-gather of a step happens after its rows are final, so every rank ends with final rows only (two gloo ranks);
  * steps that are never enqueued are gaps, not obstacles; flush() / synchronize() finalise a trailing partial group.

The stand-ins make a violation visible: the detector writes PROVISIONAL rows (negative numbers) at predict time and the final rows only at
its flush; the predictor copies what it sees of the track into its outputs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from wtracker_amd.pipeline import TrackPipeline


class _Folded:
    input_frames = [0, -2, -9, -11, -18, -20, -27]
    pred_frames = [9]


class HoldBackDetector:
    """Rows of a call are -1 (provisional) until the flush that follows `defer` calls; final row of frame f = (f, 2f, 3, 4)."""

    def __init__(self, defer: int):
        self.defer, self.device, self.max_batch = defer, 0, 1 << 20
        self._calls, self._waiting, self.flushes = 0, [], 0

    @property
    def pending(self):
        return self._calls if self.defer > 1 else 0

    def predict(self, frames_dev, B, H, W, C, out_xywh, out_conf=None, out_anchor=None, conf=0.1, iou=0.7, max_det=1, stream=0):
        ids = frames_dev.reshape(B, -1)[:, 0].to(torch.float32)  # the test encodes the global frame number in the first pixel pair
        ids = ids + 256.0 * frames_dev.reshape(B, -1)[:, 1].to(torch.float32)
        final = torch.stack([ids, 2 * ids, torch.full_like(ids, 3.0), torch.full_like(ids, 4.0)], dim=1)
        if self.defer <= 1:
            out_xywh.copy_(final)
            return
        out_xywh.fill_(-1.0)
        self._waiting.append((out_xywh, final))
        self._calls += 1
        if self._calls % self.defer == 0:
            self.flush(stream)

    def flush(self, stream=0):
        for out, final in self._waiting:
            out.copy_(final)
        self._waiting, self._calls = [], 0
        self.flushes += 1


class CopyPredictor:
    """moves[i] = (x of the anchor frame's row, x of the row 27 frames back); valid = both rows final (>= 0) and inside the track."""

    def __init__(self):
        self.launches = []

    def predict_track(self, track, total_frames, anchors, n, input_frames, moves, valid, stream=0):
        self.launches.append((int(anchors[0]), int(n)))
        for i in range(n):
            a = int(anchors[i])
            back = a + min(input_frames)
            if back < 0:
                valid[i] = 0
                moves[i] = 0
                continue
            moves[i, 0], moves[i, 1] = track[a, 0], track[back, 0]
            valid[i] = 1 if (track[a, 0] >= 0 and track[back, 0] >= 0) else -1  # -1: the predictor saw a provisional or missing row


def _frames(first, B):
    f = torch.zeros((B, 4, 4), dtype=torch.uint8)
    ids = torch.arange(first, first + B)
    f[:, 0, 0] = (ids % 256).to(torch.uint8)
    f[:, 0, 1] = (ids // 256).to(torch.uint8)
    return f


def _run(defer, lanes, steps=11, B=32, skip=(), world=1, rank=0, group=None):
    dets = [HoldBackDetector(defer) for _ in range(lanes)]
    mlp = CopyPredictor()
    pipe = TrackPipeline(dets, mlp, _Folded(), B, steps * B * world, imaging_frame_num=6, pred_frame_num=3, cycle_frame_num=9, rank=rank, world=world,
                         group=group, device=torch.device("cpu"))
    for s in range(steps):
        if s in skip:
            continue
        f0, _ = pipe.plan.local_range(s, rank)
        pipe.step(s, _frames(f0, B))
    pipe.synchronize()
    return pipe.track.numpy().copy(), pipe.moves.numpy().copy(), pipe.valid.numpy().copy(), mlp.launches, dets


@pytest.mark.parametrize("lanes", [1, 2, 3])
@pytest.mark.parametrize("defer", [2, 3, 5])
def test_deferred_rows_never_reach_the_predictor_before_they_are_final(lanes, defer):
    t1, m1, v1, l1, _ = _run(1, lanes)
    td, md, vd, ld, dets = _run(defer, lanes)
    np.testing.assert_array_equal(t1, td)
    np.testing.assert_array_equal(v1, vd)
    np.testing.assert_array_equal(m1, md)
    assert (vd >= 0).all() and vd.sum() > 20          # no cycle was predicted from a provisional row
    assert (td[:, 0] == np.arange(len(td))).all()     # every row final at the end (the trailing partial group was flushed)
    assert all(d.pending == 0 for d in dets)
    assert sum(n for _, n in ld) == sum(n for _, n in l1) == len(md)  # every cycle predicted exactly once
    assert len(ld) < len(l1)                          # ... in fewer, larger launches (one per run of steps that became final together)


def test_steps_that_are_never_enqueued_are_gaps_not_obstacles():
    t, m, v, launches, _ = _run(3, 2, skip=(4, 5))
    B = 32
    assert np.isnan(t[4 * B : 6 * B]).all() and (t[: 4 * B, 0] >= 0).all() and (t[6 * B :, 0] >= 0).all()
    # cycles whose rows (or look-back) fall into the gap come out invalid (-1 from the stand-in: it saw NaN), the others are predicted
    assert (v == 1).sum() > 10 and (v == -1).sum() > 0


def _rank_main(rank, world, port, defer, q):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = _run(defer, 2, steps=7, B=32, world=world, rank=rank)
    q.put((rank, out[0], out[1], out[2]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("defer", [1, 3])
def test_two_gloo_ranks_gather_final_rows_only(defer):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, defer, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    one = _run(1, 1, steps=7, B=64)  # one rank, the whole super-batch per step
    for _, t, m, v in res:
        np.testing.assert_array_equal(t, one[0])  # rank order == frame order, final rows only
        np.testing.assert_array_equal(v, one[2])
        np.testing.assert_array_equal(m, one[1])


def test_warning_when_more_streams_than_default_hardware_queues(monkeypatch):
    """VERDICT r04 housekeeping: a library user who keeps five streams busy without GPU_MAX_HW_QUEUES loses 3 % (or 40 % on an unlucky layout) silently."""
    import warnings

    from wtracker_amd import hip

    monkeypatch.setattr(hip, "_warned_queues", False)
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert hip.warn_if_streams_exceed_hw_queues(4) is False
        assert hip.warn_if_streams_exceed_hw_queues(5) is True
        assert hip.warn_if_streams_exceed_hw_queues(6) is False  # once per process
    assert len(w) == 1 and "request_hw_queues" in str(w[0].message)
    monkeypatch.setattr(hip, "_warned_queues", False)
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    assert hip.warn_if_streams_exceed_hw_queues(5) is False


def test_controller_picks_the_plan_by_call_size(monkeypatch, tmp_path):
    """HipYoloController's handle selection (no GPU: hip.HipYolo is replaced by a recorder).  plan "auto": the reference's single-frame call
    (provide_movement_vector, yolo_controller.py:96-98) -> a latency-plan handle, its cycle batch (_cycle_predict_all, :108-109) -> a throughput-plan
    handle, both sized for 16 frames; larger batches -> a throughput handle of YoloConfig.max_batch; fp16 never gets the latency plan; explicit plans keep
    every call of up to 16 frames on ONE handle."""
    import numpy as np

    from wtracker_amd import controllers, hip
    from wtracker_amd import yolo_spec as ys

    made = []

    class FakeYolo:
        def __init__(self, weights, imgsz, max_batch, dtype="fp16", plan="auto", **kw):
            self.max_batch, self.dtype, self.plan = max_batch, dtype, plan
            made.append((imgsz, max_batch, dtype, plan))

        def close(self):
            pass

    monkeypatch.setattr(hip, "HipYolo", FakeYolo)
    monkeypatch.delenv("WTK_LATENCY_PLAN", raising=False)
    path = str(tmp_path / "w.wtk")
    ys.save_weights(path, ys.synthetic_weights("n", 1, seed=0), "n", 1)
    mk = lambda **kw: controllers.YoloConfig(model_path=path, scale="n", **kw).load_model()
    m = mk(dtype="f16x3")
    a, b, c = m.detector((384, 384), 1), m.detector((384, 384), 15), m.detector((384, 384), 4)
    assert (a.plan, a.max_batch) == ("latency", 4) and (b.plan, b.max_batch) == ("throughput", 16) and c is a and m.detector((384, 384), 9) is b  # (auto: the latency handle is sized for the calls it sees)
    big = m.detector((384, 384), 40)
    assert (big.plan, big.max_batch) == ("throughput", 64) and big is not b
    m = mk(dtype="fp16")
    assert m.detector((384, 384), 1).plan == "throughput"
    m = mk(dtype="fp32", plan="latency")
    one = m.detector((384, 384), 1)
    assert one.plan == "latency" and m.detector((384, 384), 15) is one
    m = mk(dtype="fp32", plan="throughput")
    one = m.detector((384, 384), 15)
    assert one.plan == "throughput" and m.detector((384, 384), 1) is one
    monkeypatch.setenv("WTK_LATENCY_PLAN", "0")  # the variable turns AUTO's latency choice off
    assert mk(dtype="f16x3").detector((384, 384), 1).plan == "throughput"
