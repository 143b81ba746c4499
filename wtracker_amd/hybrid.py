"""fp16 speed with full-precision decisions: a detector pair that looks twice at the frames it is least sure about.

The fp16 detector reports a decision margin per frame (distance between its best and second-best anchor logit, or between the
best logit and the confidence threshold: wtk_yolo_margin_buffer).  Where that margin is smaller than the fp16 logit noise
(~0.01-0.02 class-logit units through 25 layers of fp16 storage) the surviving anchor can differ from the one the reference's
fp32 arithmetic picks (yolo/yolo_train_config.yaml:51 `half: False`).  HybridDetector runs the fp16 handle on the whole batch,
then — without a host round trip — the K frames with the smallest margins through a full-precision handle ("f16x3": split-fp16
operands, fp32-grade results at 2.4x the fp32 mode's rate; or "fp32"), and merges the rows whose margin is below `margin`.
K is fixed, so the step is a constant sequence of launches on one stream; the full-precision handle reads the number of weak frames
from device memory (wtk_yolo_set_dynamic_batch) and its kernels skip the tiles of the slots beyond it, so the second look costs what
the weak frames cost, not what K frames cost.

K is a CEILING on the rows that can get a second look.  The default is K = the batch size (k=None): then no weak row can be cut
off, whatever the data — 288 GB of HBM hold a full-batch f16x3 workspace easily (6.8 GB at 64 frames of 640x640) and the cost
still follows the number of weak rows.  A caller that passes a smaller K trades memory for a bound: weak rows beyond K keep their
fp16 result, and `overflow` (device counter, `overflow_count()` on the host) says how many did — it must read 0 for the
"full-precision decisions" claim to hold on that run.

How wide must `margin` be?  At least twice the fp16 logit noise of the MODEL AT HAND — and that noise is a property of the weights: with
the synthetic weight draw bench.py uses (seed 0) no fp16 survivor mismatch in thousands of frames has a margin above 0.019, with other
draws (seeds 2 and 3) mismatches reach margins of 0.10-0.15 (tests/test_gpu_hybrid_validation.py, profiles/r03_hybrid_validation.json).
A fixed number is therefore NOT a guarantee.  `calibrate()` measures, on frames of the caller's choosing, the noise of the fp16 decision margin against the full-precision one
(every frame a sample: its robust sigma and its largest absolute value) and the largest margin of any outright disagreement, and sets
margin = max(6 sigma, 1.5 x max|noise|, 2 x that largest margin, 0.02); the procedure is validated out of sample (calibration frames and validation frames disjoint) on four weight draws in the tests.  Where the calibrated margin makes most frames
weak, the hybrid is slower than the full-precision mode alone and "f16x3" is the mode to use — bench.py reports both.

It has the detector interface TrackPipeline uses (predict / predict_views / device / max_batch), so `dets=[HybridDetector(...)]`
turns the open-loop pipeline into the hybrid mode.  torch: device memory only.
"""
from __future__ import annotations

import torch

from . import hip


class HybridDetector:
    def __init__(self, fast: hip.HipYolo, exact: hip.HipYolo, margin: float = 0.04, k: int | None = None, defer: int = 1):
        if fast.device != exact.device:
            raise hip.WtkError("HybridDetector: both handles must live on the same device")
        if defer < 1:
            raise hip.WtkError("HybridDetector: defer >= 1")
        self.defer = int(defer)
        if k is None:
            k = exact.max_batch if defer > 1 else min(fast.max_batch, exact.max_batch)
        if k < 1 or k > exact.max_batch:
            raise hip.WtkError("HybridDetector: 1 <= k <= max_batch of the full-precision handle")
        self.fast, self.exact, self.margin, self.k = fast, exact, float(margin), int(k)
        self.device, self.max_batch = fast.device, fast.max_batch
        self.dtype = f"{fast.dtype}+{exact.dtype}"
        self.macs_per_frame, self.anchors = fast.macs_per_frame, fast.anchors
        dev = torch.device("cuda", self.device)
        self._slots = torch.zeros((self.k,), dtype=torch.int32, device=dev)
        self._xywh = torch.empty((self.k, 4), dtype=torch.float32, device=dev)
        self._conf = torch.empty((self.k,), dtype=torch.float32, device=dev)
        self._anchor = torch.empty((self.k,), dtype=torch.int32, device=dev)
        self._pos = None  # view centre that makes a "view" the whole frame, per frame shape (defer = 1) / enqueue scratch (defer > 1)
        self._idx_tmp = torch.empty((self.k,), dtype=torch.int32, device=dev)
        self._pos_tmp = torch.empty((self.k, 2), dtype=torch.int32, device=dev)
        self.replaced = torch.zeros((1,), dtype=torch.int32, device=dev)  # rows replaced so far (device counter)
        self._n_weak = torch.zeros((1,), dtype=torch.int32, device=dev)  # weak rows of the current batch: the second look's dynamic batch size
        self.overflow = torch.zeros((1,), dtype=torch.int32, device=dev)  # weak rows so far that the ceiling k cut off (0 by construction when k >= B)
        exact.set_dynamic_batch(self._n_weak)
        # ---- deferred mode (defer = D > 1): the weak rows of D consecutive calls share ONE full-precision pass (wtk_recheck_enqueue /
        # wtk_recheck_scatter): `_n_weak` is then the length of a device-side queue of frame copies + output-row addresses.  The pass has a
        # fixed cost of ~1.2 ms however few frames are live (62 launches of a few tiles each); per call that is 1.2 ms / D instead of 1.2 ms.
        # With k >= D x (fast batch) no weak row can find the queue full.  A smaller queue (the full-precision pass launches its grids for k frames
        # and the blocks of slots beyond the queue's length exit at once, ~5 us per unused slot: k should not be much larger than what is
        # expected) COUNTS what it has to drop (`overflow`): such rows keep their fp16 result, and a caller that promises full-precision
        # decisions must check the counter (bench.py does, and demotes the mode if it is not 0).
        # Rows named in a call are FINAL only after the flush that follows (every D-th call, or flush()): `pending` says how many calls wait.
        self._calls = 0
        self._q_frames = None
        self._q_shape = None
        if self.defer > 1:
            self._q_ptrs = [torch.zeros((self.k,), dtype=torch.int64, device=dev) for _ in range(3)]
            self._pos = torch.empty((max(fast.max_batch, 1),), dtype=torch.int32, device=dev)
        # (the second look keeps the default concurrency: with ONE pair of side streams per process, shared by every handle, its towers cost no
        # extra streams — 17.7 k frames/s against 17.1 k with wtk_yolo_set_side_streams(0); with a pair per handle it was the other way round)

    # -- the detector interface -------------------------------------------------------------------------------------------
    def predict(self, frames_dev, B: int, H: int, W: int, Cc: int, out_xywh, out_conf=None, out_anchor=None, conf: float = 0.1,
                iou: float = 0.7, max_det: int = 1, stream: int = 0):
        if max_det != 1:
            raise hip.WtkError("HybridDetector: max_det must be 1")
        self.fast.predict(frames_dev, B, H, W, Cc, out_xywh, out_conf, out_anchor, conf=conf, iou=iou, max_det=1, stream=stream)
        if self.defer > 1:
            if self._q_frames is None:
                self._q_shape = (H, W, Cc)
                self._q_frames = torch.empty((self.k, H, W, Cc), dtype=torch.uint8, device=self._slots.device)
            elif self._q_shape != (H, W, Cc):
                raise hip.WtkError("HybridDetector(defer > 1): every call must bring frames of the same shape")
            if (H * W * Cc) % 16:
                raise hip.WtkError("HybridDetector(defer > 1): frames must be a multiple of 16 bytes")
            hip.recheck_enqueue(self.fast.margin_buffer(), B, self.margin, frames_dev, H * W * Cc, self._q_frames, self.k, self._n_weak, *self._q_ptrs,
                                out_xywh, out_conf, out_anchor, self._pos, self.overflow, stream=stream)
            self._calls += 1
            self._conf_thr = conf
            if self._calls % self.defer == 0:
                self.flush(stream)
            return
        k = min(self.k, B)
        m = self.fast.margin_buffer()
        hip.recheck_select(m, B, k, self.margin, self._slots, self._n_weak, stream=stream, n_overflow_dev=self.overflow)
        if self._pos is None or self._pos[0] != (H, W):
            # wtk_yolo_predict_views cuts frame[y0 : y0 + view_w, x0 : x0 + view_h] with (x0, y0) = pos - (view_w // 2, view_h // 2)
            # (view_controller.py:158-172): view (H, W) centred there is the frame itself
            p = torch.tensor([[H // 2, W // 2]] * self.k, dtype=torch.int32).to(self._slots.device)
            self._pos = ((H, W), p)
        self.exact.predict_views(frames_dev, B, H, W, Cc, self._slots, self._pos[1], k, H, W, self._xywh, self._conf, self._anchor, conf=conf, iou=iou,
                                 max_det=1, stream=stream)
        hip.recheck_merge(m, self._slots, B, k, self.margin, self._xywh, self._conf, self._anchor, out_xywh, out_conf, out_anchor, self.replaced, stream=stream)

    @property
    def pending(self) -> int:
        """Calls whose weak rows still wait for their full-precision pass (deferred mode; 0 = every row handed out so far is final)."""
        return self._calls if self.defer > 1 else 0

    def flush(self, stream: int = 0):
        """Deferred mode: look again at everything queued so far and write the rows back (enqueued on `stream`, no host synchronisation)."""
        if self.defer <= 1 or self._q_frames is None or self._calls == 0:
            return
        H, W, Cc = self._q_shape
        self.exact.predict(self._q_frames, self.k, H, W, Cc, self._xywh, self._conf, self._anchor, conf=getattr(self, "_conf_thr", 0.1), max_det=1, stream=stream)
        hip.recheck_scatter(self._n_weak, self.k, self._xywh, self._conf, self._anchor, *self._q_ptrs, self.replaced, stream=stream)
        self._calls = 0

    def predict_views(self, frames_dev, n_frames: int, H: int, W: int, Cc: int, frame_index_dev, pos_xy_dev, B: int, view_w: int, view_h: int,
                      out_xywh, out_conf=None, out_anchor=None, conf: float = 0.1, iou: float = 0.7, max_det: int = 1, stream: int = 0):
        if max_det != 1:
            raise hip.WtkError("HybridDetector: max_det must be 1")
        if self.defer > 1:
            raise hip.WtkError("HybridDetector(defer > 1): the views entry point has no deferred form (use defer = 1)")
        self.fast.predict_views(frames_dev, n_frames, H, W, Cc, frame_index_dev, pos_xy_dev, B, view_w, view_h, out_xywh, out_conf, out_anchor, conf=conf,
                                iou=iou, max_det=1, stream=stream)
        k = min(self.k, B)
        m = self.fast.margin_buffer()
        hip.recheck_select(m, B, k, self.margin, self._slots, self._n_weak, stream=stream, n_overflow_dev=self.overflow)
        # the weak rows' (frame, position): gathered by torch on the caller's stream (it must be torch's current stream)
        sl = self._slots[:k].long()
        if frame_index_dev is None:
            self._idx_tmp[:k] = self._slots[:k]
        else:
            torch.index_select(frame_index_dev, 0, sl, out=self._idx_tmp[:k])
        torch.index_select(pos_xy_dev, 0, sl, out=self._pos_tmp[:k])
        self.exact.predict_views(frames_dev, n_frames, H, W, Cc, self._idx_tmp, self._pos_tmp, k, view_w, view_h, self._xywh, self._conf, self._anchor,
                                 conf=conf, iou=iou, max_det=1, stream=stream)
        hip.recheck_merge(m, self._slots, B, k, self.margin, self._xywh, self._conf, self._anchor, out_xywh, out_conf, out_anchor, self.replaced, stream=stream)

    def calibrate(self, batches, H: int, W: int, Cc: int = 1, conf: float = 0.1, safety: float = 2.0, z: float = 6.0, floor: float = 0.02, tail: float = 1.5) -> dict:
        """Set `margin` from measurements on THIS model: every batch of `batches` (device uint8 tensors [B, H, W(, C)], B <= max_batch of both
        handles) goes through the fast and through the full-precision handle.  Two statistics bound how far fp16 can move a decision:
          * the NOISE of the decision margin itself, d = margin(fast) - margin(full precision) on the frames where both name the same survivor
            (every frame is a sample): sigma = max(std, 1.4826 MAD, p99 / 2.576) of d.  A survivor can only flip where the full-precision gap
            between two anchors is smaller than fp16's perturbation of that gap, and the flipped frame's fp16 margin is at most that perturbation;
          * the largest |d| seen: the noise is not Gaussian for every model — one weight draw of the validation shows |d| of 10 sigma on 512 frames and,
            in 16 384 frames, a disagreement with a margin of 5.9 sigma — so the tail gets its own term;
          * the largest fast-pass margin of any frame on which the two DO disagree (few samples: only the tail).
        margin = max(floor, z x sigma, tail x max|d|, safety x largest mismatch margin).  Returns what was measured.  Synchronises; not for timed regions."""
        if self.exact.max_batch < self.fast.max_batch:
            raise hip.WtkError("calibrate: the full-precision handle must take whole batches (max_batch >= the fast handle's)")
        import numpy as np

        self.exact.set_dynamic_batch(None)
        dev = torch.device("cuda", self.device)
        worst, n_frames, n_bad, margins_bad, noise, all_margins = 0.0, 0, 0, [], [], []
        try:
            for fb in batches:
                B = int(fb.shape[0])
                o = [(torch.empty((B, 4), dtype=torch.float32, device=dev), torch.empty((B,), dtype=torch.float32, device=dev),
                      torch.empty((B,), dtype=torch.int32, device=dev)) for _ in range(2)]
                self.fast.predict(fb, B, H, W, Cc, *o[0], conf=conf)
                self.exact.predict(fb, B, H, W, Cc, *o[1], conf=conf)
                torch.cuda.synchronize(dev)
                m, mx = self.fast.last_margins(B), self.exact.last_margins(B)
                bad = (o[0][2] != o[1][2]).cpu().numpy()
                all_margins.extend(float(v) for v in m)
                n_frames += B
                n_bad += int(bad.sum())
                d = (m - mx)[~bad]
                noise.extend(float(v) for v in d[np.isfinite(d)])
                if bad.any():
                    margins_bad.extend(float(v) for v in m[bad])
                    worst = max(worst, float(np.nanmax(m[bad])))
        finally:
            self.exact.set_dynamic_batch(self._n_weak)
        noise = np.asarray(noise, dtype=np.float64)
        if len(noise) >= 8:
            mad = float(np.median(np.abs(noise - np.median(noise))))
            sigma = max(float(noise.std()), 1.4826 * mad, float(np.percentile(np.abs(noise), 99)) / 2.576)
        else:
            sigma = 0.0
        d_max = float(np.abs(noise).max()) if len(noise) else 0.0
        self.margin = max(float(floor), float(z) * sigma, float(tail) * d_max, float(safety) * worst)
        share = float((np.asarray(all_margins) < self.margin).mean()) if all_margins else 0.0
        return {"frames": n_frames, "share_below_margin": share, "fast_mismatches": n_bad, "largest_mismatch_margin": worst, "margin_noise_sigma": sigma, "margin_noise_max_abs": d_max,
                "safety": safety, "z": z, "tail": tail, "floor": floor, "margin": self.margin, "mismatch_margins_sorted_desc": sorted(margins_bad, reverse=True)[:8]}

    def overflow_count(self) -> int:
        """Weak rows (margin below the threshold) that kept their fp16 result because more than k rows of a batch were weak.  Synchronises."""
        return int(self.overflow.item())

    def set_profiling(self, enabled: bool):
        self.fast.set_profiling(enabled)

    def get_profile(self) -> dict:
        return self.fast.get_profile()

    def get_kernel_profile(self) -> dict:
        return self.fast.get_kernel_profile()

    def close(self):
        self.fast.close()
        self.exact.set_dynamic_batch(None)
        self.exact.close()
