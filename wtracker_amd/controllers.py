"""Drop-in controllers: the reference's `SimController` implementations for the hot path, with the
arithmetic executed by libwtk_hip.so on the MI355X.

Mirrors (same constructor arguments, method names, return conventions, error behaviour):
  CsvController      wtracker/sim/sim_controllers/csv_controller.py:11-73
  OptimalController  wtracker/sim/sim_controllers/optimal_controller.py:8-32   (host arithmetic)
  PolyfitConfig / PolyfitController  wtracker/sim/sim_controllers/polyfit_controller.py:13-84  (host arithmetic, f4)
  HipOptimalController / HipPolyfitController  the same two with all cycles evaluated on the device (track_ops.hip, f4)
  HipMLPController   <- MLPController   wtracker/sim/sim_controllers/mlp_controllers.py:14-71
  YoloConfig         wtracker/sim/sim_controllers/yolo_controller.py:15-45
  HipYoloController  <- YoloController  wtracker/sim/sim_controllers/yolo_controller.py:48-109
"""
from __future__ import annotations

import os
from collections import deque
from dataclasses import dataclass, field
from typing import Any, Collection, Optional

import numpy as np

from . import hip, yolo_spec
from .resmlp import FoldedResMLP, from_torch_module
from .sim import SimController, TimingConfig


def _read_track_csv(path: str) -> np.ndarray:
    """Columns wrm_x, wrm_y, wrm_w, wrm_h as float64 [N,4]; empty cells -> NaN.  Parsed with pandas'
    default C float parser on purpose: it is what the reference uses (csv_controller.py:16) and it is
    not round-trip exact, so any other parser differs from the reference's table in the last ulp."""
    import pandas as pd

    return pd.read_csv(path, usecols=["wrm_x", "wrm_y", "wrm_w", "wrm_h"]).to_numpy(dtype=float)


class _CameraRing:
    """The camera boxes (x, y, w, h) of the most recent `capacity` camera frames, in one preallocated float64 array written
    round-robin.  `oldest_first(i)` is the i-th oldest box still held — the element order of the bounded queue the reference
    keeps (csv_controller.py:18,23) — and asking for a position that is not held yet is an IndexError, as it is there."""

    def __init__(self, capacity: int):
        self.capacity = int(capacity)
        self.boxes = np.zeros((self.capacity, 4), dtype=np.float64)
        self.pushed = 0  # boxes written since the last reset

    def reset(self):
        self.pushed = 0

    def push(self, box):
        self.boxes[self.pushed % self.capacity] = box
        self.pushed += 1

    def oldest_first(self, positions: np.ndarray) -> np.ndarray:
        held = min(self.pushed, self.capacity)
        if positions.size and (int(positions.max()) >= held or int(positions.min()) < 0):
            raise IndexError(f"camera history holds {held} frames; position {int(positions.max())} requested")
        return self.boxes[(self.pushed - held + positions) % self.capacity]


class CsvController(SimController):
    """Replays pre-detected head boxes from a bboxes.csv log: the record/replay feeder under the ResMLP, polynomial and look-ahead
    controllers.  Same constructor, methods and results as csv_controller.py:11-73 (pinned by the real reference's logs,
    tests/test_sim_golden.py), stated over two small structures of this module: the track as a table with one extra all-NaN row
    that every out-of-range frame number is mapped to, and the camera history as a round-robin array (_CameraRing)."""

    def __init__(self, timing_config: TimingConfig, csv_path: str):
        super().__init__(timing_config)
        self.csv_path = csv_path
        self.track = _read_track_csv(csv_path)                       # [N,4] float64 xywh, NaN row = missed detection
        self._table = np.vstack([self.track, np.full((1, 4), np.nan)])  # row N: what a frame outside the log reads
        self._cameras = _CameraRing(timing_config.cycle_frame_num)

    @property
    def n_frames(self) -> int:
        return self.track.shape[0]

    def on_sim_start(self, sim):
        self._cameras.reset()

    def on_camera_frame(self, sim):
        self._cameras.push(sim.view.camera_position)

    def predict(self, frame_nums: Collection[int], relative: bool = True) -> np.ndarray:
        """Boxes of the given frames, [k,4] float64 (fresh array); frames outside the log are NaN rows.  relative=True: x, y are
        measured from the corner of the camera view of that frame's slot in the current cycle (csv_controller.py:39-47: only
        meaningful for frames of the cycle being replayed — the reference's own TODO)."""
        assert len(frame_nums) > 0
        wanted = np.asanyarray(frame_nums, dtype=int)
        inside = (wanted >= 0) & (wanted < self.n_frames)
        rows = self._table[np.where(inside, wanted, self.n_frames)]
        if relative:
            rows[:, :2] -= self._cameras.oldest_first(wanted % self.timing_config.cycle_frame_num)[:, :2]
        return rows

    def begin_movement_prediction(self, sim) -> None:
        pass

    def provide_movement_vector(self, sim) -> tuple:
        """Centre the camera on the head as it was seen `pred_frame_num` frames ago (csv_controller.py:54-68); (0, 0) when that frame has no box."""
        x, y, w, h = self.predict([sim.frame_number - self.timing_config.pred_frame_num])[0]
        if not (np.isfinite(x) and np.isfinite(y) and np.isfinite(w) and np.isfinite(h)):
            return 0, 0
        cam_w, cam_h = sim.view.camera_size
        return round((x + w / 2) - cam_w / 2), round((y + h / 2) - cam_h / 2)  # Python's banker's rounding on float64, as the reference

    def _cycle_predict_all(self, sim) -> np.ndarray:
        """The finished cycle's boxes relative to their camera views (what LoggingController logs, logging_controller.py:145-154)."""
        n = self.timing_config.cycle_frame_num
        first = (sim.cycle_number - 1) * n
        return self.predict(np.arange(first, min(first + n, self.n_frames)))


class OptimalController(CsvController):
    """Upper-bound baseline: looks into the future of the replayed track and centres the camera on the
    median head position of the NEXT imaging phase (optimal_controller.py:8-32).  A few flops per cycle:
    stays on the host."""

    def __init__(self, timing_config: TimingConfig, csv_path: str):
        super().__init__(timing_config, csv_path)
        d = self.track
        self._csv_centers = np.stack([d[:, 0] + d[:, 2] / 2, d[:, 1] + d[:, 3] / 2], axis=1)

    def provide_movement_vector(self, sim) -> tuple:
        lo = (sim.cycle_number + 1) * self.timing_config.cycle_frame_num
        window = self._csv_centers[lo : lo + self.timing_config.imaging_frame_num, :]
        window = window[np.isfinite(window).all(axis=1)]
        if len(window) == 0:
            return 0, 0
        x_next, y_next = np.median(window, axis=0)
        cx, cy, cw, ch = sim.view.camera_position
        return round(x_next - (cx + cw / 2)), round(y_next - (cy + ch / 2))


@dataclass
class PolyfitConfig:
    """degree / sample_times / weights of the weighted polynomial fit (polyfit_controller.py:13-27).
    sample_times are frame offsets from the START of the current cycle."""

    degree: int
    sample_times: list
    weights: Optional[list] = None

    def __post_init__(self):
        # the reference sorts the times but NOT the weights (polyfit_controller.py:28): weight i belongs to the i-th
        # smallest time, whatever order the caller wrote the times in
        self.sample_times = sorted(self.sample_times)
        if self.weights is None:
            self.weights = [1.0 for _ in self.sample_times]
        assert len(self.sample_times) == len(self.weights)


class PolyfitController(CsvController):
    """Classical baseline the ResMLP is compared with: weighted least-squares polynomial through the
    sampled head centres, evaluated at the middle of the next imaging phase
    (polyfit_controller.py:30-84).  Uses numpy.polynomial.polynomial like the reference so the
    coefficients are the same bits."""

    def __init__(self, timing_config: TimingConfig, polyfit_config: PolyfitConfig, csv_path: str) -> None:
        super().__init__(timing_config, csv_path)
        self.polyfit_config = polyfit_config
        self._sample_times = np.asanyarray(polyfit_config.sample_times, dtype=int)
        self._weights = np.asanyarray(polyfit_config.weights, dtype=float)

    def provide_movement_vector(self, sim) -> tuple:
        from numpy.polynomial import polynomial as poly

        timing = self.timing_config
        boxes = self.predict(sim.cycle_number * timing.cycle_frame_num + self._sample_times, relative=False)
        cam = sim.view.camera_position
        boxes[:, 0] -= cam[0]
        boxes[:, 1] -= cam[1]
        centers = np.stack([boxes[:, 0] + boxes[:, 2] / 2, boxes[:, 1] + boxes[:, 3] / 2], axis=1)
        ok = np.isfinite(centers).all(axis=1)
        t = self._sample_times[ok]
        if len(t) == 0:
            return 0, 0
        coeffs = poly.polyfit(t, centers[ok], deg=self.polyfit_config.degree, w=self._weights[ok])
        x_pred, y_pred = poly.polyval(timing.cycle_frame_num + timing.imaging_frame_num // 2, coeffs)
        return round(x_pred - sim.view.camera_size[0] / 2), round(y_pred - sim.view.camera_size[1] / 2)


class _DeviceTrackMixin:
    """Uploads the replayed track (float64 [N,4], NaN rows = missed detections) once and evaluates a per-cycle predictor for
    ALL cycles of the experiment in one launch (libwtk_hip.so, track_ops.hip).  The predictors' targets do not depend on the
    platform position — only the final `round(target - camera centre)` does, and that stays on the host in float64."""

    def _upload_track(self, device: int):
        import torch

        self._dev = torch.device("cuda", device)
        self._track_dev = torch.from_numpy(np.ascontiguousarray(self.track, dtype=np.float64)).to(self._dev)
        n_cycles = self.n_frames // self.timing_config.cycle_frame_num + 2
        self._cycles_dev = torch.arange(n_cycles, dtype=torch.int32, device=self._dev)
        self._pred_dev = torch.zeros((n_cycles, 2), dtype=torch.float64, device=self._dev)
        self._valid_dev = torch.zeros((n_cycles,), dtype=torch.int32, device=self._dev)
        return n_cycles

    def _download(self):
        import torch

        torch.cuda.synchronize(self._dev)
        return self._pred_dev.cpu().numpy(), self._valid_dev.cpu().numpy().astype(bool)


class HipOptimalController(_DeviceTrackMixin, CsvController):
    """OptimalController (optimal_controller.py:8-32) with the per-cycle medians of the whole track computed on the device
    (wtk_track_median_centers) when the controller is built."""

    def __init__(self, timing_config: TimingConfig, csv_path: str, device: int = 0):
        CsvController.__init__(self, timing_config, csv_path)
        import torch

        n = self._upload_track(device)
        with torch.cuda.device(self._dev):
            hip.track_median_centers(self._track_dev, self.n_frames, self._cycles_dev, n, timing_config.cycle_frame_num,
                                     timing_config.imaging_frame_num, self._pred_dev, self._valid_dev,
                                     stream=torch.cuda.current_stream(self._dev).cuda_stream)
        self._targets, self._has_target = self._download()

    def provide_movement_vector(self, sim) -> tuple:
        c = sim.cycle_number
        if c >= len(self._targets) or not self._has_target[c]:
            return 0, 0
        x_next, y_next = self._targets[c]
        cx, cy, cw, ch = sim.view.camera_position
        return round(x_next - (cx + cw / 2)), round(y_next - (cy + ch / 2))


class HipPolyfitController(_DeviceTrackMixin, CsvController):
    """PolyfitController (polyfit_controller.py:35-84) with the weighted fits of all cycles computed on the device
    (wtk_track_polyfit).  The reference fits camera-relative centres and subtracts the camera half size; the fit commutes
    with that translation, so the device fits absolute centres and the host subtracts the camera centre (float64).  The device
    solver agrees with numpy's LAPACK lstsq to ~1e-10 px, far inside the integer rounding of the move."""

    def __init__(self, timing_config: TimingConfig, polyfit_config: PolyfitConfig, csv_path: str, device: int = 0):
        CsvController.__init__(self, timing_config, csv_path)
        import torch

        self.polyfit_config = polyfit_config
        n = self._upload_track(device)
        t_eval = timing_config.cycle_frame_num + timing_config.imaging_frame_num // 2
        with torch.cuda.device(self._dev):
            hip.track_polyfit(self._track_dev, self.n_frames, self._cycles_dev, n, timing_config.cycle_frame_num,
                              polyfit_config.sample_times, polyfit_config.weights, polyfit_config.degree, t_eval, self._pred_dev,
                              self._valid_dev, stream=torch.cuda.current_stream(self._dev).cuda_stream)
        self._targets, self._has_target = self._download()

    def provide_movement_vector(self, sim) -> tuple:
        c = sim.cycle_number
        if c >= len(self._targets) or not self._has_target[c]:
            return 0, 0
        x_pred, y_pred = self._targets[c]
        cx, cy, cw, ch = sim.view.camera_position
        # reference: (x_pred - cam_x) - cam_w / 2 on camera-relative fits
        return round((x_pred - cx) - sim.view.camera_size[0] / 2), round((y_pred - cy) - sim.view.camera_size[1] / 2)


class HipMLPController(CsvController):
    """MLPController with the ResMLP forward on the GPU (wtk_mlp_forward_host).

    `model` may be a `FoldedResMLP` (see wtracker_amd.resmlp) or a live reference `WormPredictor`
    (anything with `state_dict()` and `io_config`), as simulate.ipynb cell 9 passes."""

    def __init__(self, timing_config: TimingConfig, csv_path: str, model, max_speed: float = 0.9, device: int = 0):
        super().__init__(timing_config, csv_path)
        folded: FoldedResMLP = model if isinstance(model, FoldedResMLP) else from_torch_module(model)
        self.model = folded
        self.input_frames = list(folded.input_frames)
        self.pred_frames = list(folded.pred_frames)
        self._mlp = hip.HipMLP(folded.layers, folded.n_blocks, folded.layers_per_block, device=device)
        max_speed_px_frame = max_speed * (timing_config.px_per_mm / timing_config.frames_per_sec)
        self.max_dist_per_pred = max_speed_px_frame * self.pred_frames[0]

    def provide_movement_vector(self, sim) -> tuple:
        frames = np.asanyarray(self.input_frames, dtype=int) + (sim.frame_number - self.timing_config.pred_frame_num)
        cam = np.asanyarray(sim.view.camera_position)
        cam_center = (cam[0] + cam[2] / 2, cam[1] + cam[3] / 2)
        boxes = self.predict(frames, relative=False).reshape(1, -1)  # float64, NaN when out of range
        if not np.isfinite(boxes).all():
            return 0, 0
        # the model is fed the box CORNER (not centre) of frame 0 as origin (mlp_controllers.py:46-56)
        x0, y0 = boxes[0, 0], boxes[0, 1]
        rel_x, rel_y = x0 - cam_center[0], y0 - cam_center[1]
        boxes[:, 0::4] -= x0
        boxes[:, 1::4] -= y0
        pred = self._mlp.forward_host(boxes.astype(np.float32))[0]  # float64 -> float32 as torch.Tensor(ndarray) does
        pred = np.clip(pred, -self.max_dist_per_pred, self.max_dist_per_pred)
        # float64 add + Python banker's rounding stay on the host, exactly as written in the reference
        return round(pred[0].item() + rel_x), round(pred[1].item() + rel_y)


@dataclass
class YoloConfig:
    """Same fields as the reference's YoloConfig.  `model_path` is a WTKYOLO1 weight file
    (wtracker_amd.yolo_spec.save_weights); `device` 'cuda' / 'cuda:N' selects the MI355X."""

    model_path: str
    device: str = "cuda"
    verbose: bool = False
    pred_kwargs: dict = field(default_factory=lambda: {"imgsz": 384, "conf": 0.1})
    # extensions (not in the reference): arithmetic type and model scale.  fp32 (exact-fp32 MFMA) is the default because the
    # reference computes in fp32 (yolo/yolo_train_config.yaml:51 `half: False`): survivor indices equal the fp32 restatement's.
    # 'fp16' is the opt-in throughput mode: 6.5x faster, ~97 % survivor-index match, IoU >= 0.997 on matched frames
    # (tests/test_gpu_configs.py::test_fp16_accuracy_vs_fp32_oracle_at_640_b64).
    # 'f16x3' (YOLOv8 s / l): the same results as 'fp32' (every conv tensor within 4e-6 of its scale, survivor index equal) from
    # three fp16 MFMAs per product on split operands, 2.4x faster; activations must stay below the fp16 maximum (65504).
    # 'auto' = 'f16x3' where the scale has it, else 'fp32'.
    dtype: str = "fp32"
    # fp16 mode only: frames whose decision margin (best vs second-best anchor logit, or best logit vs the conf threshold; class-logit
    # units, wtk_yolo_last_margins_host) is below this value are detected AGAIN by a full-precision handle, whose result replaces the row.
    # The fp16 head logits are within ~0.01-0.02 of the fp32 ones, so a margin of 0.08 restores the fp32 restatement's survivor
    # on every frame of the 256-frame accuracy set (tests/test_gpu_configs.py) while ~20 % of the frames are re-run.  0 = off.
    recheck_margin: float = 0.0
    # precision of that second look: "auto" = "f16x3" where the scale has it (s, l: fp32-grade results at 2.4x the fp32 mode's rate,
    # tests/test_gpu_f16x3.py), else "fp32"
    recheck_dtype: str = "auto"
    scale: str = "s"
    max_batch: int = 64
    # launch plan of the detector handles (include/wtk_hip.h: wtk_yolo_create_planned).  The reference calls the detector twice per cycle: one single frame
    # (provide_movement_vector, yolo_controller.py:96-98) and one cycle batch of 9 / 15 frames (_cycle_predict_all, :108-109).  "auto": calls of up to
    # LATENCY_MAX_BATCH frames go to a handle on the latency plan (split-K convs, replayed captures: 0.53 ms instead of 1.0 ms for the single frame), larger
    # ones to a throughput-plan handle (1.2 ms instead of 1.5 ms for 15 frames): each call on the plan that is faster for it; the two handles agree within
    # the tolerance both meet against the fp32 restatement, not bit for bit.  "latency": every call up to 16 frames on ONE latency-plan handle — a frame's
    # result is then bit-identical whichever of the two calls sees it; "throughput": every call on the large-batch kernels.  fp16: always throughput.
    plan: str = "auto"
    model: Any = field(default=None, init=False, repr=False)

    def __getstate__(self) -> dict:
        state = self.__dict__.copy()
        del state["model"]  # like the reference, never serialise the model (yolo_controller.py:37-40)
        return state

    def device_index(self) -> int:
        if self.device == "cpu":
            raise hip.WtkError("HipYoloController has no CPU path: use device='cuda' (the MI355X)")
        return int(self.device.split(":")[1]) if ":" in self.device else 0

    def recheck_mode(self, nc: int = 1) -> str:
        if self.recheck_dtype != "auto":
            return self.recheck_dtype
        return "f16x3" if yolo_spec.split_capable(self.scale, nc) else "fp32"

    def load_model(self) -> "_YoloModel":
        if self.model is None:
            self.model = _YoloModel(self)
        return self.model


def _raise_on_overflow(det: hip.HipYolo, flags: Optional[int] = None) -> None:
    """The fp16-storage modes need |activation| < 65 504 (include/wtk_hip.h: wtk_yolo_status).  A trained, BatchNorm-folded checkpoint
    (yolo_controller.py:42-45 loads one) whose activations leave that range would otherwise give NaN rows or wrong survivors silently: the head
    kernels raise a sticky flag when a head logit is inf / NaN, and the controller turns it into an error that names the way out.
    `flags`: the status word as it was read when the call finished (lanes that share a handle: HipYoloController.launch_views); None = read it now."""
    if flags is None:
        flags = det.status(clear=True)
    if flags & hip.STATUS_NONFINITE:
        raise hip.WtkError(f"non-finite head logits in dtype '{det.dtype}': an activation of this model left the fp16 range (65 504); "
                           "use YoloConfig(dtype='fp32')" if det.dtype != "fp32" else "non-finite head logits: the model's weights or the input produce inf / NaN")


class _YoloModel:
    """Weights + one device detector per network input shape (created on first use)."""

    def __init__(self, cfg: YoloConfig):
        self.cfg = cfg
        self.weights, self.nc = yolo_spec.load_weights(cfg.model_path)
        self._dets: dict = {}
        self._retired: list = []

    LATENCY_MAX_BATCH = 4  # measured cross-over between the plans near B = 6 at imgsz 384 (profiles/r05_notes.md)

    def _retire(self, det: hip.HipYolo) -> None:
        """A handle that is replaced by a larger one: a call launched on it (HipYoloController.launch_views, lane 1) may still be running and its token
        still names the handle — the device drains before the handle goes, and the handle stays readable (status) until its tokens are collected."""
        import torch

        if torch.cuda.is_available():
            torch.cuda.synchronize(self.cfg.device_index())
        self._retired.append(det)  # closed when the model goes: collect_views may still ask it for its status word

    def detector(self, net_hw: tuple, batch: int, dtype: Optional[str] = None) -> hip.HipYolo:
        dtype = dtype or self.cfg.dtype
        if dtype == "auto":  # the reference's precision at the best rate this scale has
            dtype = "f16x3" if yolo_spec.split_capable(self.cfg.scale, self.nc) else "fp32"
        plan = self.cfg.plan
        if plan == "auto" or dtype in ("fp16", "f16", "float16"):
            plan = "latency" if batch <= self.LATENCY_MAX_BATCH and dtype not in ("fp16", "f16", "float16") and os.environ.get("WTK_LATENCY_PLAN") != "0" else "throughput"
        key = (net_hw, dtype, plan)
        det = self._dets.get(key)
        if det is None or det.max_batch < batch:
            if det is not None:
                self._retire(det)
            width, depth, maxch = yolo_spec.scale_params(self.cfg.scale)
            # sizes: a latency-plan handle under plan "auto" only ever sees calls of up to LATENCY_MAX_BATCH frames (its split-K slabs are sized by the
            # capacity: 4, not 16); a throughput-plan handle for the reference's own calls (<= 16 frames) is sized for 16 — the library runs the
            # 12 x 12-map layers of such a handle on its split-K kernel (csrc/wtk_plan.hip: sk_mixed); plan "latency" keeps both calls of a cycle on one
            # handle of 16
            if plan == "latency" and self.cfg.plan == "auto":
                cap = self.LATENCY_MAX_BATCH
            else:
                cap = 16 if batch <= 16 else max(batch, self.cfg.max_batch)
            det = hip.HipYolo(self.weights, net_hw, cap, dtype=dtype, nc=self.nc, width=width, depth=depth, max_channels=maxch,
                              device=self.cfg.device_index(), plan=plan)
            self._dets[key] = det
        return det


class HipYoloController(SimController):
    """YoloController with the detector on the MI355X.

    `device_frames` (extension): a torch uint8 CUDA tensor [F,H,W] or [F,H,W,3] holding the experiment's
    full frames (what the sim's frame reader serves, already resident in HBM).  The controller then never calls
    `sim.camera_view()`: per camera frame it records (frame number, platform position), and the camera views are cut
    (replicate border) and letterboxed ON THE DEVICE in front of the detector (wtk_yolo_predict_views) — no host crop, no
    per-cycle upload.  Without it the controller behaves exactly like the reference: host crops through `predict(frames)`."""

    def __init__(self, timing_config: TimingConfig, yolo_config: YoloConfig, device_frames=None):
        super().__init__(timing_config)
        self.yolo_config = yolo_config
        self._camera_frames = deque(maxlen=timing_config.cycle_frame_num)
        self._model = yolo_config.load_model()
        self._device_frames = device_frames
        self.last_rechecked = 0  # frames of the last predict call that were detected again at full precision (YoloConfig.recheck_margin)
        self._view_streams: dict = {}  # device-resident path: the controller's own stream per lane + buffers per (lane, batch size) (launch_views)
        self._view_bufs: dict = {}
        self._inflight: dict = {}  # id(detector handle) -> token of the call that may still run on it
        if device_frames is not None:
            if not getattr(device_frames, "is_cuda", False) or str(device_frames.dtype) != "torch.uint8" or device_frames.dim() not in (3, 4):
                raise hip.WtkError("device_frames must be a CUDA uint8 tensor [F,H,W] or [F,H,W,3]")
            if not device_frames.is_contiguous():
                raise hip.WtkError("device_frames must be contiguous")

    def on_sim_start(self, sim):
        self._camera_frames.clear()

    def on_camera_frame(self, sim):
        if self._device_frames is not None:  # (frame number, platform position): 12 bytes instead of a w x h crop
            self._camera_frames.append((int(sim.view.index), int(sim.view.position[0]), int(sim.view.position[1]), tuple(sim.view.camera_size)))
        else:
            self._camera_frames.append(sim.camera_view())

    def on_cycle_end(self, sim):
        self._camera_frames.clear()

    def predict(self, frames: Collection[np.ndarray]) -> np.ndarray:
        """frames: HxW (gray) or HxWx3 (BGR) uint8 arrays of one shape -> [N,4] (x, y, w, h) in frame
        pixels; a frame without a detection above `conf` yields a NaN row.  float32, or float64 when
        any row is NaN (np.stack of float32 boxes and float64 NaN rows, yolo_controller.py:85-90)."""
        assert len(frames) > 0
        batch = np.stack([np.asarray(f) for f in frames], axis=0)
        if batch.dtype != np.uint8:
            raise hip.WtkError("frames must be uint8")
        kw = dict(self.yolo_config.pred_kwargs)
        imgsz = int(kw.pop("imgsz", 640))
        conf = float(kw.pop("conf", 0.25))
        iou = float(kw.pop("iou", 0.7))
        if "max_det" in kw:
            raise TypeError("predict() got multiple values for keyword argument 'max_det'")  # as the reference would
        H, W = batch.shape[1], batch.shape[2]
        net_hw = yolo_spec.letterbox_shape(H, W, imgsz)
        det = self._model.detector(net_hw, batch.shape[0])
        xywh, _, anchor = det.predict_host(batch, conf=conf, iou=iou, max_det=1)
        _raise_on_overflow(det)
        self.last_rechecked = 0
        if self.yolo_config.recheck_margin > 0 and det.dtype in ("fp16", "f16", "float16"):
            weak = np.nonzero(det.last_margins(batch.shape[0]) < self.yolo_config.recheck_margin)[0]
            if len(weak):  # ambiguous frames: the reference's precision decides
                x32, _, a32 = self._model.detector(net_hw, len(weak), dtype=self.yolo_config.recheck_mode(self._model.nc)).predict_host(
                    batch[weak], conf=conf, iou=iou, max_det=1)
                xywh[weak], anchor[weak] = x32, a32
                self.last_rechecked = len(weak)
        if (anchor < 0).any():
            out = xywh.astype(np.float64)
            out[anchor < 0] = np.nan
            return out
        return xywh

    def predict_views(self, entries) -> np.ndarray:
        """`entries`: (frame number, position x, position y, (view_w, view_h)) tuples recorded by on_camera_frame.  Same return
        convention as predict(): [N,4] xywh in VIEW pixels, NaN rows for frames without a detection."""
        return self.collect_views(self.launch_views(entries))

    def launch_views(self, entries, lane: int = 0) -> dict:
        """First half of predict_views: enqueue the view cut + letterbox + detector for `entries` on the controller's stream of `lane` and return a
        token for collect_views().  Nothing is waited for.  Lane 1 is what the deferred track log uses for the cycle batch: it runs beside the next
        cycle's single-frame call (lane 0) instead of in front of it.  Tokens of one lane must be collected before that lane launches again."""
        import torch

        assert len(entries) > 0
        fr = self._device_frames
        kw = dict(self.yolo_config.pred_kwargs)
        imgsz, conf, iou = int(kw.pop("imgsz", 640)), float(kw.pop("conf", 0.25)), float(kw.pop("iou", 0.7))
        if "max_det" in kw:
            raise TypeError("predict() got multiple values for keyword argument 'max_det'")
        vw, vh = entries[0][3]
        n = len(entries)
        net_hw = yolo_spec.letterbox_shape(vw, vh, imgsz)  # the view's shape is (rows = w, cols = h)
        det = self._model.detector(net_hw, n)
        dev = fr.device
        # One set of device buffers per (lane, batch size), kept for the controller's life: the view table (frame numbers + positions, one small upload
        # per call from pinned memory) and the output rows.  The library replays a captured forward pass for a call that comes back with the same
        # device addresses (wtk_yolo_predict_views: the reference's operating point — one cycle batch and one single-frame call per cycle — is
        # launch bound), which needs a stream of the controller's own: the legacy default stream cannot be captured.
        stream = self._view_streams.get(lane)
        if stream is None:
            # (a high-priority stream for lane 0 — the call the loop waits for — changes nothing once the handles run on one stream each: deferred log 11.19 k
            # against 10.81-11.19 k frames/s, single-frame call 0.87 against 0.86 ms; with side streams it cost the cycle batch 30 %: profiles/r06_notes.md section 4)
            stream = self._view_streams[lane] = torch.cuda.Stream(device=dev)
        # A detector handle has ONE set of activation buffers: a call on another lane that still runs on the same handle (plan "latency" / "throughput":
        # both calls of a cycle share a handle) is waited for first; with plan "auto" the two calls have a handle each and overlap.
        busy = self._inflight.get(id(det))
        if busy is not None and busy["lane"] != lane:
            busy["stream"].synchronize()
            # the handle's sticky status word now holds THAT call's flags: they go into its token before this call can raise its own (a flag is
            # reported by the call that caused it, whichever lane collects first)
            busy["flags"] = det.status(clear=True)
        # EVERY call is ordered behind the caller's current stream: a caller that refills or extends `device_frames` in place between cycles has its
        # writes on that stream, and the crop / letterbox kernel must not read frames that are still being written.  The order costs an event record +
        # wait (~90 us of host time in front of the first launch when the current stream is the legacy default stream: cProfile, round 6) only when
        # that stream HAS work pending — an idle stream (hipStreamQuery: microseconds) has nothing to be ordered behind.
        cur = torch.cuda.current_stream(dev)
        if os.environ.get("WTK_CTRL_ALWAYS_WAIT") == "1" or not cur.query():
            stream.wait_stream(cur)
        bufs = self._view_bufs.get((lane, n))
        if bufs is None:
            bufs = self._view_bufs[(lane, n)] = dict(meta=torch.empty((3 * n,), dtype=torch.int32, device=dev), host=torch.empty((3 * n,), dtype=torch.int32).pin_memory(),
                                                     out=torch.empty((n, 4), dtype=torch.float32, device=dev), cf=torch.empty((n,), dtype=torch.float32, device=dev),
                                                     an=torch.empty((n,), dtype=torch.int32, device=dev),
                                                     # the rows come back through pinned memory: two asynchronous copies behind the forward pass on the same stream and
                                                     # ONE synchronisation (two blocking `.cpu()` calls cost ~40 us of a 0.5 ms call)
                                                     out_h=torch.empty((n, 4), dtype=torch.float32).pin_memory(), an_h=torch.empty((n,), dtype=torch.int32).pin_memory())
            bufs["host_np"] = bufs["host"].numpy()
        hn = bufs["host_np"]
        hn[:n] = [e[0] for e in entries]
        hn[n:] = [v for e in entries for v in (e[1], e[2])]
        host = bufs["host"]
        meta, out, cf, an = bufs["meta"], bufs["out"], bufs["cf"], bufs["an"]
        idx, pos = meta[:n], meta[n:].view(n, 2)
        C = fr.shape[3] if fr.dim() == 4 else 1
        with torch.cuda.device(dev), torch.cuda.stream(stream):
            meta.copy_(host, non_blocking=True)
            det.predict_views(fr, fr.shape[0], fr.shape[1], fr.shape[2], C, idx, pos, n, vw, vh, out, cf, an, conf=conf, iou=iou, max_det=1, stream=stream.cuda_stream)
            bufs["out_h"].copy_(out, non_blocking=True)
            bufs["an_h"].copy_(an, non_blocking=True)
        token = dict(lane=lane, stream=stream, det=det, bufs=bufs, n=n, view=(vw, vh), net_hw=net_hw, conf=conf, iou=iou, C=C)
        self._inflight[id(det)] = token
        return token

    def collect_views(self, token: dict) -> np.ndarray:
        """Second half of predict_views: wait for the token's call and return its rows."""
        import torch

        stream, det, bufs, n = token["stream"], token["det"], token["bufs"], token["n"]
        fr = self._device_frames
        dev = fr.device
        stream.synchronize()
        if self._inflight.get(id(det)) is token:
            del self._inflight[id(det)]
        xywh, anchor = bufs["out_h"].numpy().copy(), bufs["an_h"].numpy().copy()
        _raise_on_overflow(det, token.get("flags"))
        self.last_rechecked = 0
        if self.yolo_config.recheck_margin > 0 and det.dtype in ("fp16", "f16", "float16"):
            weak = np.nonzero(det.last_margins(n) < self.yolo_config.recheck_margin)[0]
            if len(weak):
                vw, vh = token["view"]
                out, cf, an = bufs["out"], bufs["cf"], bufs["an"]
                idx, pos = bufs["meta"][:n], bufs["meta"][n:].view(n, 2)
                wsel = torch.from_numpy(weak).to(dev)
                det32 = self._model.detector(token["net_hw"], len(weak), dtype=self.yolo_config.recheck_mode(self._model.nc))
                k = len(weak)
                with torch.cuda.device(dev), torch.cuda.stream(stream):
                    det32.predict_views(fr, fr.shape[0], fr.shape[1], fr.shape[2], token["C"], idx[wsel].contiguous(), pos[wsel].contiguous(), k, vw, vh,
                                        out[:k], cf[:k], an[:k], conf=token["conf"], iou=token["iou"], max_det=1, stream=stream.cuda_stream)
                    xywh[weak], anchor[weak] = out[:k].cpu().numpy(), an[:k].cpu().numpy()
                self.last_rechecked = k
        if (anchor < 0).any():
            res = xywh.astype(np.float64)
            res[anchor < 0] = np.nan
            return res
        return xywh

    def begin_movement_prediction(self, sim) -> None:
        pass

    def provide_movement_vector(self, sim) -> tuple:
        frame = self._camera_frames[-self.timing_config.pred_frame_num]
        bbox = (self.predict_views([frame]) if self._device_frames is not None else self.predict([frame]))[0]
        if not np.isfinite(bbox).all():
            return 0, 0
        mid = bbox[0] + bbox[2] / 2, bbox[1] + bbox[3] / 2
        cam_mid = sim.view.camera_size[0] / 2, sim.view.camera_size[1] / 2
        return round(mid[0] - cam_mid[0]), round(mid[1] - cam_mid[1])

    def _cycle_predict_all(self, sim) -> np.ndarray:
        if self._device_frames is not None:
            return self.predict_views(list(self._camera_frames))
        return self.predict(self._camera_frames)

    def _cycle_predict_all_async(self, sim):
        """The cycle batch enqueued on lane 1 (device-resident frames only; else None): TrackLogger(deferred=True) collects it one cycle later, so the
        batch runs on the GPU beside the next cycle's single-frame call instead of in front of it.  Same rows as _cycle_predict_all."""
        if self._device_frames is None or self.yolo_config.recheck_margin > 0:  # (the second look reads the handle's margins of its LAST call: immediate log)
            return None
        return self.launch_views(list(self._camera_frames), lane=1)

    def _cycle_collect(self, token) -> np.ndarray:
        return self.collect_views(token)
