"""Open-loop batched form of the sim loop's hot path (BASELINE configs 3-5): frames that are
already camera views are detected in batches (the `_cycle_predict_all` batch call,
yolo_controller.py:108-109, widened from a 9/15-frame cycle to a 64/256-frame super-batch), the
resulting bbox track is exchanged between ranks, and the ResMLP movement prediction of every
imaging/moving cycle whose prediction frame falls in the super-batch is computed from the track
(`MLPController.provide_movement_vector`, mlp_controllers.py:36-68).

Multi-GPU: frames are sharded over ranks in interleaved super-batches — step s covers frames
[s*B*N, (s+1)*B*N), rank r detects [s*B*N + r*B, +B) — so after ONE all-gather per step (RCCL over
xGMI, payload B*16 bytes per rank: latency-bound) every rank holds the whole track up to the end of
the super-batch, which is what ResMLP's 27..45-frame look-back needs.  No other
collective exists on the path.

torch is used for device memory, streams and torch.distributed only; all arithmetic is in
libwtk_hip.so.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import hip
from .resmlp import FoldedResMLP


class ShardPlan:
    """Pure index arithmetic of the interleaved super-batch sharding (no device work): which frames a
    rank detects at step s, and which cycles' predictions become computable after that step."""

    def __init__(self, batch: int, world: int, total_frames: int, imaging_frame_num: int, pred_frame_num: int, cycle_frame_num: int):
        self.B, self.world = batch, world
        self.super_batch = batch * world
        self.total_frames = total_frames
        self.img, self.pred, self.cyc = imaging_frame_num, pred_frame_num, cycle_frame_num
        n_cycles = max(0, (total_frames - imaging_frame_num) // cycle_frame_num + 1)
        anchors = np.arange(n_cycles, dtype=np.int64) * cycle_frame_num + (imaging_frame_num - pred_frame_num)
        self.anchors = anchors[anchors + pred_frame_num < total_frames].astype(np.int32)

    @property
    def steps(self) -> int:
        return self.total_frames // self.super_batch

    def local_range(self, s: int, rank: int) -> tuple:
        f0 = s * self.super_batch + rank * self.B
        return f0, f0 + self.B

    def super_range(self, s: int) -> tuple:
        return s * self.super_batch, (s + 1) * self.super_batch

    def cycles(self, s: int) -> tuple:
        """[lo, hi) into self.anchors: cycles whose provide_movement_vector frame lies in super-batch s."""
        f0, f1 = self.super_range(s)
        key = self.anchors.astype(np.int64) + self.pred
        return int(np.searchsorted(key, f0, side="left")), int(np.searchsorted(key, f1, side="left"))


def exchange_tracks(track: torch.Tensor, local_xywh: torch.Tensor, plan: ShardPlan, s: int, group=None, comm=None, stream: int = 0):
    """The path's only collective: all-gather the ranks' [B,4] slices into the track rows of super-batch s
    (rank order == frame order, so the gathered block is contiguous).  Through torch.distributed (RCCL on the GPUs, gloo in
    the CPU tests) or, with `comm` (hip.WtkComm), through the C ABI's own RCCL communicator (wtk_allgather_tracks)."""
    f0, f1 = plan.super_range(s)
    if comm is not None:
        comm.allgather_tracks(local_xywh, plan.B, track[f0:f1], stream=stream)
        return
    import torch.distributed as dist

    dist.all_gather_into_tensor(track[f0:f1].view(-1), local_xywh.contiguous().view(-1), group=group)


class TrackPipeline:
    """`dets`: one detector handle per lane.  With two lanes consecutive super-batches run on two HIP streams
    (each with its own activation workspace), so the tails and the latency-bound layers of one forward pass
    are filled by the other; the ResMLP of step s waits for the detections of steps s and s-1 (its 27-frame
    look-back never reaches further than one super-batch of >= 27 frames)."""

    def __init__(self, dets, mlp: hip.HipMLP, folded: FoldedResMLP, batch: int, total_frames: int,
                 imaging_frame_num: int, pred_frame_num: int, cycle_frame_num: int, conf: float = 0.1,
                 rank: int = 0, world: int = 1, group=None, device: Optional[torch.device] = None, comm=None, streams=None):
        self.dets = list(dets) if isinstance(dets, (list, tuple)) else [dets]
        self.mlp, self.folded = mlp, folded
        self.rank, self.world, self.group, self.comm = rank, world, group, comm
        self.conf = conf
        self.device = device or torch.device("cuda", self.dets[0].device)
        self._cuda = self.device.type == "cuda"  # a CPU device runs the same schedule without streams / events (tests/test_pipeline_schedule.py drives it with stand-in detectors)
        self.plan = ShardPlan(batch, world, total_frames, imaging_frame_num, pred_frame_num, cycle_frame_num)
        lookback = -min(folded.input_frames) + pred_frame_num
        if len(self.dets) > 1 and self.plan.super_batch < lookback:
            raise ValueError("two lanes need super-batches of at least the predictor's look-back")
        n_lanes = len(self.dets)
        # `streams`: one torch stream per lane from the caller (a process that builds several pipelines keeps the same lane streams: torch hands
        # out pooled streams round-robin, and which hardware queue a stream sits on changes what runs concurrently with what)
        if streams is not None and len(streams) != n_lanes:
            raise ValueError("one stream per lane")
        if self._cuda and n_lanes > 1:  # lanes + the process-wide pair of side streams + the caller's default stream
            hip.warn_if_streams_exceed_hw_queues(n_lanes + 2 + 1)
        self.streams = list(streams) if streams is not None else ([torch.cuda.Stream(device=self.device) for _ in range(n_lanes)] if n_lanes > 1 and self._cuda else [None] * n_lanes)
        # A lane whose detector holds rows back (HybridDetector(defer = D): the weak rows of D calls share one full-precision pass) hands out
        # rows that are FINAL only after its next flush.  Everything downstream of the rows — the exchange between ranks and the ResMLP —
        # runs when a lane's rows become final, for every step that is final on ALL lanes by then; a detector without `defer` is final at
        # once and the schedule is the plain one (detect, exchange, predict, per step).
        self._ring = [max(int(getattr(d, "defer", 1)), 1) for d in self.dets]
        self.final_ev = [torch.cuda.Event() if self._cuda else None for _ in range(n_lanes)]
        self.det_done = self.final_ev  # (older name)
        # device-resident track of the whole run: xywh per frame (NaN = no detection yet / none found)
        self.track = torch.full((total_frames, 4), float("nan"), dtype=torch.float32, device=self.device)
        mk = lambda shape, dt: torch.empty(shape, dtype=dt, device=self.device)
        self.local_xywh = [[mk((batch, 4), torch.float32) for _ in range(r)] for r in self._ring]  # one set per step a lane may hold back
        self.local_conf = [[mk((batch,), torch.float32) for _ in range(r)] for r in self._ring]
        self.local_anchor = [[mk((batch,), torch.int32) for _ in range(r)] for r in self._ring]
        n = max(len(self.plan.anchors), 1)
        self.anchors = torch.from_numpy(self.plan.anchors).to(self.device)
        self.moves = torch.zeros((n, 2), dtype=torch.float32, device=self.device)  # raw ResMLP (dx, dy) per cycle
        self.valid = torch.zeros((n,), dtype=torch.int32, device=self.device)
        self._steps_done = 0
        self._calls = [0] * n_lanes            # detector calls per lane (ring position)
        self._held = [[] for _ in range(n_lanes)]  # (step, ring slot) of a lane's steps whose rows are not final yet
        self._ready = {}                       # step -> lane: rows final, ResMLP of its cycles not enqueued yet

    def _finalize_lane(self, lane: int) -> int:
        """The rows of every step this lane holds are final in the order of the current stream (= the lane's): exchange them between the
        ranks, then run the ResMLP for the cycles of every step that is now final on all lanes.  Returns the cycles launched."""
        cur = torch.cuda.current_stream(self.device) if self._cuda else None
        st = cur.cuda_stream if cur is not None else 0
        for t, k in self._held[lane]:
            if self.world > 1:
                exchange_tracks(self.track, self.local_xywh[lane][k], self.plan, t, self.group, self.comm, st)
            self._ready[t] = lane
        self._held[lane] = []
        if cur is not None:
            self.final_ev[lane].record(cur)
        # A final step can be processed once no lane still holds an earlier step back (the ResMLP of step t looks back into step t - 1);
        # steps that were never enqueued are gaps, not obstacles (their rows stay NaN and the cycles that need them come out invalid).
        first_held = min((t for h in self._held for t, _ in h), default=None)
        todo = sorted(t for t in self._ready if first_held is None or t < first_held)
        if not todo:
            return 0
        for other in range(len(self.dets)):  # the look-back of these steps reaches into rows that other lanes finalised on their own streams
            if other != lane and self._calls[other] > 0 and cur is not None:
                cur.wait_event(self.final_ev[other])
        n, i = 0, 0
        while i < len(todo):  # one launch per run of consecutive steps (their cycles are contiguous)
            j = i
            while j + 1 < len(todo) and todo[j + 1] == todo[j] + 1:
                j += 1
            lo, hi = self.plan.cycles(todo[i])[0], self.plan.cycles(todo[j])[1]
            if hi > lo:
                self.mlp.predict_track(self.track, self.plan.total_frames, self.anchors[lo:hi], hi - lo, self.folded.input_frames,
                                       self.moves[lo:hi], self.valid[lo:hi], stream=st)
                n += hi - lo
            i = j + 1
        for t in todo:
            del self._ready[t]
        return n

    def _step_on_current_stream(self, s: int, lane: int, frames_dev: torch.Tensor, views=None) -> int:
        H, W = frames_dev.shape[1], frames_dev.shape[2]
        C = frames_dev.shape[3] if frames_dev.dim() == 4 else 1
        det = self.dets[lane]
        st = torch.cuda.current_stream(self.device).cuda_stream if self._cuda else 0
        f0, f1 = self.plan.local_range(s, 0)
        k = self._calls[lane] % self._ring[lane]
        self._calls[lane] += 1
        out = self.track[f0:f1] if self.world == 1 else self.local_xywh[lane][k]  # one rank: straight into the track
        if views is None:
            det.predict(frames_dev, frames_dev.shape[0], H, W, C, out, self.local_conf[lane][k], self.local_anchor[lane][k], conf=self.conf, stream=st)
        else:  # camera views of full frames: crop + letterbox on the device in front of the detector
            frame_index, pos_xy, (vw, vh) = views
            det.predict_views(frames_dev, frames_dev.shape[0], H, W, C, frame_index, pos_xy, self.plan.B, vw, vh, out, self.local_conf[lane][k],
                              self.local_anchor[lane][k], conf=self.conf, stream=st)
        self._held[lane].append((s, k))
        self._steps_done += 1
        if getattr(det, "pending", 0) == 0:
            return self._finalize_lane(lane)
        return 0

    def step(self, s: int, frames_dev: torch.Tensor, views=None) -> int:
        """Detect this rank's B frames of super-batch s, exchange, predict the super-batch's cycles (at once, or — with a detector that holds
        rows back — when the rows become final: flush() / synchronize() make everything final).
        `views` = (frame_index [B] int32 or None, pos_xy [B,2] int32, (view_w, view_h)): `frames_dev` then holds FULL frames
        and the batch rows are their camera views (device crop + letterbox).
        Returns the number of cycles whose ResMLP was enqueued by THIS call: with a plain detector the cycles of step s; with a deferring
        detector (HybridDetector(defer = D)) 0 for the D - 1 steps a lane holds back and the cycles of all of them at the lane's flush —
        movement vectors are then available up to D x lanes super-batches after their frames were seen."""
        lane = s % len(self.dets)
        if self.streams[lane] is None:
            return self._step_on_current_stream(s, lane, frames_dev, views)
        stream = self.streams[lane]
        stream.wait_stream(torch.cuda.current_stream(self.device))  # inputs produced on the caller's stream
        with torch.cuda.stream(stream):
            return self._step_on_current_stream(s, lane, frames_dev, views)

    def flush(self) -> int:
        """Make every row handed out so far final (detectors that hold rows back look again at what they queued) and run what waited for
        them.  Enqueues work on the lanes' streams; no host synchronisation."""
        n = 0
        for lane, det in enumerate(self.dets):
            if not self._held[lane]:
                continue
            stream = self.streams[lane]
            if stream is None:
                if getattr(det, "pending", 0):
                    det.flush(torch.cuda.current_stream(self.device).cuda_stream if self._cuda else 0)
                n += self._finalize_lane(lane)
            else:
                with torch.cuda.stream(stream):
                    if getattr(det, "pending", 0):
                        det.flush(stream.cuda_stream)
                    n += self._finalize_lane(lane)
        return n

    def baseline_targets(self, sample_times=None, weights=None, degree: int = 2):
        """The other two predictors of the reference over the finished device track, every cycle in one launch each and no
        host pass: OptimalController's median head position of the NEXT imaging phase
        (optimal_controller.py:16-32) and, when `sample_times` is given, PolyfitController's weighted-fit extrapolation
        (polyfit_controller.py:54-84).  Returns {name: (targets float64 [n_cycles,2] absolute px, valid int32 [n_cycles])} as
        device tensors; cycle i is the cycle of self.plan.anchors[i].  Call after the last step; rows a deferring detector still holds
        back are made final first (flush()), so the targets never read a row that would still change."""
        self.flush()
        n = len(self.plan.anchors)
        st = torch.cuda.current_stream(self.device)
        for ev in self.det_done:
            if ev is not None:
                st.wait_event(ev)
        cycles = torch.arange(n, dtype=torch.int32, device=self.device)
        out = {}
        tgt = torch.zeros((n, 2), dtype=torch.float64, device=self.device)
        ok = torch.zeros((n,), dtype=torch.int32, device=self.device)
        hip.track_median_centers(self.track, self.plan.total_frames, cycles, n, self.plan.cyc, self.plan.img, tgt, ok, stream=st.cuda_stream)
        out["optimal"] = (tgt, ok)
        if sample_times is not None:
            w = [1.0] * len(sample_times) if weights is None else list(weights)
            tgt2 = torch.zeros((n, 2), dtype=torch.float64, device=self.device)
            ok2 = torch.zeros((n,), dtype=torch.int32, device=self.device)
            hip.track_polyfit(self.track, self.plan.total_frames, cycles, n, self.plan.cyc, sorted(sample_times), w, degree,
                              self.plan.cyc + self.plan.img // 2, tgt2, ok2, stream=st.cuda_stream)
            out["polyfit"] = (tgt2, ok2)
        return out

    def synchronize(self):
        self.flush()
        for st in self.streams:
            if st is not None:
                st.synchronize()
