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


class _Counter:
    """A device counter of the native object read through wtk_hybrid_counters (synchronises): .item() / int()."""

    def __init__(self, read):
        self._read = read

    def item(self) -> int:
        return self._read()

    def __int__(self) -> int:
        return self._read()


class HybridDetector:
    """Thin shell over the library's wtk_hybrid object (include/wtk_hip.h): the orchestration — fast pass, slot selection or queue,
    second look with a device-side dynamic batch, merge / scatter, counters — runs behind the C ABI; what stays here is the detector
    interface TrackPipeline duck-types and `calibrate()` (host-side statistics over two passes)."""

    def __init__(self, fast: hip.HipYolo, exact: hip.HipYolo, margin: float = 0.04, k: int | None = None, defer: int = 1):
        import ctypes as C

        if fast.device != exact.device:
            raise hip.WtkError("HybridDetector: both handles must live on the same device")
        if defer < 1:
            raise hip.WtkError("HybridDetector: defer >= 1")
        lib = hip.load()
        self._h = C.c_void_p()
        hip._check(lib.wtk_hybrid_create(C.byref(self._h), fast._h, exact._h, float(margin), 0 if k is None else int(k), int(defer)), "wtk_hybrid_create")
        kk, dd = C.c_int32(), C.c_int32()
        hip._check(lib.wtk_hybrid_config(self._h, C.byref(kk), C.byref(dd), None), "wtk_hybrid_config")
        self.fast, self.exact, self.k, self.defer = fast, exact, int(kk.value), int(dd.value)
        self._margin = float(margin)
        self.device, self.max_batch = fast.device, fast.max_batch
        self.dtype = f"{fast.dtype}+{exact.dtype}"
        self.macs_per_frame, self.anchors = fast.macs_per_frame, fast.anchors
        # the native object borrows the two handles: closing either of them first releases it (HipYolo.close)
        for d in (fast, exact):
            d.__dict__.setdefault("_dependents", []).append(self)
        self.replaced = _Counter(lambda: self._counters()[0])  # rows replaced so far
        self.overflow = _Counter(lambda: self._counters()[1])  # weak rows so far that the ceiling k cut off (0 by construction when k covers every call of a flush group)

    # the threshold lives in the native object
    @property
    def margin(self) -> float:
        return self._margin

    @margin.setter
    def margin(self, value: float):
        hip._check(hip.load().wtk_hybrid_set_margin(self._h, float(value)), "wtk_hybrid_set_margin")
        self._margin = float(value)

    def _counters(self):
        import ctypes as C

        r, o = C.c_int64(), C.c_int64()
        hip._check(hip.load().wtk_hybrid_counters(self._h, C.byref(r), C.byref(o)), "wtk_hybrid_counters")
        return int(r.value), int(o.value)

    # -- the detector interface -------------------------------------------------------------------------------------------
    def predict(self, frames_dev, B: int, H: int, W: int, Cc: int, out_xywh, out_conf=None, out_anchor=None, conf: float = 0.1,
                iou: float = 0.7, max_det: int = 1, stream: int = 0):
        import ctypes as C

        if max_det != 1:
            raise hip.WtkError("HybridDetector: max_det must be 1")
        hip._check(hip.load().wtk_hybrid_predict(self._h, hip._ptr(frames_dev), B, H, W, Cc, conf, hip._ptr(out_xywh), hip._ptr(out_conf), hip._ptr(out_anchor),
                                                 C.c_void_p(stream)), "wtk_hybrid_predict")

    @property
    def pending(self) -> int:
        """Calls whose weak rows still wait for their full-precision pass (deferred mode; 0 = every row handed out so far is final)."""
        return int(hip.load().wtk_hybrid_pending(self._h))

    def flush(self, stream: int = 0):
        """Deferred mode: look again at everything queued so far and write the rows back (enqueued on `stream`, no host synchronisation)."""
        import ctypes as C

        hip._check(hip.load().wtk_hybrid_flush(self._h, C.c_void_p(stream)), "wtk_hybrid_flush")

    def predict_views(self, frames_dev, n_frames: int, H: int, W: int, Cc: int, frame_index_dev, pos_xy_dev, B: int, view_w: int, view_h: int,
                      out_xywh, out_conf=None, out_anchor=None, conf: float = 0.1, iou: float = 0.7, max_det: int = 1, stream: int = 0):
        import ctypes as C

        if max_det != 1:
            raise hip.WtkError("HybridDetector: max_det must be 1")
        hip._check(hip.load().wtk_hybrid_predict_views(self._h, hip._ptr(frames_dev), n_frames, H, W, Cc, hip._ptr(frame_index_dev), hip._ptr(pos_xy_dev), B, view_w,
                                                       view_h, conf, hip._ptr(out_xywh), hip._ptr(out_conf), hip._ptr(out_anchor), C.c_void_p(stream)),
                   "wtk_hybrid_predict_views")

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

        self._suspend()  # the full-precision handle takes whole batches while it is measured
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
            self._resume()
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

    def _suspend(self):
        """The full-precision handle takes whole batches while it is measured (wtk_hybrid_hold)."""
        hip._check(hip.load().wtk_hybrid_hold(self._h, 1), "wtk_hybrid_hold")

    def _resume(self):
        hip._check(hip.load().wtk_hybrid_hold(self._h, 0), "wtk_hybrid_hold")

    def overflow_count(self) -> int:
        """Weak rows (margin below the threshold) that kept their fp16 result because more than k rows of a batch were weak.  Synchronises."""
        return int(self.overflow.item())

    def set_profiling(self, enabled: bool):
        self.fast.set_profiling(enabled)

    def get_profile(self) -> dict:
        return self.fast.get_profile()

    def get_kernel_profile(self) -> dict:
        return self.fast.get_kernel_profile()

    def _release(self):
        if getattr(self, "_h", None) is not None and self._h:
            hip.load().wtk_hybrid_destroy(self._h)  # hands the full-precision handle its static batch back
        self._h = None

    def close(self):
        self._release()
        self.fast.close()
        self.exact.close()

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass
