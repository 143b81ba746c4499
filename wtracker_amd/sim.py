"""Host-side types the hot path's controllers are written against (SURVEY.md §8 b, f2):
  * TimingConfig / ExperimentConfig arithmetic           wtracker/sim/config.py:41-67, 105-111
  * the SimController plugin interface (hook names)      wtracker/sim/simulator.py:197-293
  * the per-cycle track log (CSV schema) + box utilities wtracker/sim/sim_controllers/logging_controller.py:95-116,145-185,
                                                         wtracker/utils/bbox_utils.py:118-167,232-253
The DRIVER that calls these hooks is the reference's own `Simulator` (a controller of this package runs inside it unchanged:
tests/test_dropin_reference.py).  The stand-in driver the tests and bench.py use where the reference is absent (frame loop, platform
motor, camera views) is test infrastructure and lives in tests/harness/sim_harness.py, not in this package (round 4).
"""
from __future__ import annotations

import abc
import csv
import math
from collections import deque
from dataclasses import dataclass
from typing import Optional

import numpy as np


# -------------------------------------------------------------------------------------------------
# configs
# -------------------------------------------------------------------------------------------------
@dataclass
class ExperimentConfig:
    name: str
    num_frames: int
    frames_per_sec: float
    orig_resolution: tuple  # (h, w)
    px_per_mm: float
    init_position: tuple  # (x, y)
    comments: str = ""

    def __post_init__(self):
        self.ms_per_frame = 1000 / self.frames_per_sec
        self.mm_per_px = 1 / self.px_per_mm

    @classmethod
    def from_dict(cls, d: dict) -> "ExperimentConfig":
        keys = ("name", "num_frames", "frames_per_sec", "orig_resolution", "px_per_mm", "init_position", "comments")
        kw = {k: d[k] for k in keys if k in d}
        kw["orig_resolution"] = tuple(kw["orig_resolution"])
        kw["init_position"] = tuple(kw["init_position"])
        return cls(**kw)


class TimingConfig:
    """ms -> frame counts by ceil, mm -> px by Python round (config.py:46-62)."""

    def __init__(self, experiment_config: ExperimentConfig, imaging_time_ms: float, pred_time_ms: float, moving_time_ms: float,
                 camera_size_mm: tuple, micro_size_mm: tuple):
        ec = experiment_config
        self.frames_per_sec = ec.frames_per_sec
        self.ms_per_frame = ec.ms_per_frame
        self.imaging_time_ms, self.pred_time_ms, self.moving_time_ms = imaging_time_ms, pred_time_ms, moving_time_ms
        self.imaging_frame_num = math.ceil(imaging_time_ms / self.ms_per_frame)
        self.pred_frame_num = math.ceil(pred_time_ms / self.ms_per_frame)
        self.moving_frame_num = math.ceil(moving_time_ms / self.ms_per_frame)
        self.mm_per_px, self.px_per_mm = ec.mm_per_px, ec.px_per_mm
        self.camera_size_mm, self.micro_size_mm = tuple(camera_size_mm), tuple(micro_size_mm)
        self.camera_size_px = (round(self.px_per_mm * camera_size_mm[0]), round(self.px_per_mm * camera_size_mm[1]))
        self.micro_size_px = (round(self.px_per_mm * micro_size_mm[0]), round(self.px_per_mm * micro_size_mm[1]))

    @property
    def cycle_frame_num(self) -> int:
        return self.imaging_frame_num + self.moving_frame_num

    @property
    def cycle_time_ms(self) -> float:
        return self.cycle_frame_num * self.ms_per_frame


# -------------------------------------------------------------------------------------------------
# plugin interface
# -------------------------------------------------------------------------------------------------
class SimController(abc.ABC):
    """Same hook names / signatures as the reference ABC (simulator.py:197-293)."""

    def __init__(self, timing_config: TimingConfig):
        self.timing_config = timing_config

    def on_sim_start(self, sim): pass
    def on_sim_end(self, sim): pass
    def on_cycle_start(self, sim): pass
    def on_cycle_end(self, sim): pass
    def on_camera_frame(self, sim): pass
    def on_imaging_start(self, sim): pass
    def on_micro_frame(self, sim): pass
    def on_imaging_end(self, sim): pass
    def on_movement_start(self, sim): pass
    def on_movement_end(self, sim): pass

    @abc.abstractmethod
    def begin_movement_prediction(self, sim) -> None: ...

    @abc.abstractmethod
    def provide_movement_vector(self, sim) -> tuple: ...

    @abc.abstractmethod
    def _cycle_predict_all(self, sim) -> np.ndarray: ...


# -------------------------------------------------------------------------------------------------
# track log (the on-disk wire format of tracks: bboxes.csv / init_bboxes.csv)
# -------------------------------------------------------------------------------------------------
LOG_COLUMNS = ["frame", "cycle", "phase", "plt_x", "plt_y", "cam_x", "cam_y", "cam_w", "cam_h",
               "mic_x", "mic_y", "mic_w", "mic_h", "wrm_x", "wrm_y", "wrm_w", "wrm_h"]


class TrackLogger(SimController):
    """Decorator controller: forwards every hook and, at each cycle end, asks the wrapped controller
    for the whole cycle's boxes (`_cycle_predict_all` -> the batched detector call), makes them
    absolute and appends one CSV row per frame.  Rows are also kept in `self.rows`."""

    def __init__(self, sim_controller: SimController, csv_path: Optional[str] = None, deferred: bool = False):
        """`deferred` (extension): with a controller that offers `_cycle_predict_all_async` / `_cycle_collect` (HipYoloController on device-resident
        frames) the cycle batch is only ENQUEUED at the cycle's end and its rows are written one cycle later (the last cycle's at on_sim_end): the
        batch — which feeds nothing back into the loop — then runs on the GPU beside the next cycle's single-frame call.  Same rows, same file."""
        super().__init__(sim_controller.timing_config)
        self.sim_controller = sim_controller
        self.csv_path = csv_path
        self.deferred = bool(deferred)
        self._pending = None  # (token, cycle, platform / camera / micro positions) of the cycle batch that is still running
        n = self.timing_config.cycle_frame_num
        self._plt, self._cam, self._mic = deque(maxlen=n), deque(maxlen=n), deque(maxlen=n)
        self.rows: list = []
        self._file = None
        self._writer = None

    def on_sim_start(self, sim):
        self.sim_controller.on_sim_start(sim)
        self._plt.clear(), self._cam.clear(), self._mic.clear()
        self.rows = []
        self._pending = None
        if self.csv_path:
            self._file = open(self.csv_path, "w+", newline="")
            self._writer = csv.DictWriter(self._file, LOG_COLUMNS, escapechar=",")
            self._writer.writeheader()

    def on_camera_frame(self, sim):
        self.sim_controller.on_camera_frame(sim)
        self._plt.append(sim.position)
        self._cam.append(sim.view.camera_position)
        self._mic.append(sim.view.micro_position)

    def on_cycle_end(self, sim):
        cycle = sim.cycle_number - 1
        launch = getattr(self.sim_controller, "_cycle_predict_all_async", None) if self.deferred else None
        if launch is not None:
            self._flush_pending()  # the previous cycle's batch has had a whole cycle to finish
            token = launch(sim)
            if token is not None:
                self._pending = (token, cycle, list(self._plt), list(self._cam), list(self._mic))
                self.sim_controller.on_cycle_end(sim)
                self._plt.clear(), self._cam.clear(), self._mic.clear()
                return
        self._write_rows(self.sim_controller._cycle_predict_all(sim), cycle, self._plt, self._cam, self._mic)
        self.sim_controller.on_cycle_end(sim)
        self._plt.clear(), self._cam.clear(), self._mic.clear()

    def _flush_pending(self):
        if self._pending is not None:
            token, cycle, plt, cam, mic = self._pending
            self._pending = None
            self._write_rows(self.sim_controller._cycle_collect(token), cycle, plt, cam, mic)

    def _write_rows(self, boxes, cycle, plt, cam, mic):
        first = cycle * self.timing_config.cycle_frame_num
        cams = np.asanyarray(list(cam))
        boxes[:, 0] += cams[:, 0]
        boxes[:, 1] += cams[:, 1]
        # The reference calls BoxUtils.discretize on this array before writing the rows, and that
        # function zeroes non-finite rows IN PLACE (bbox_utils.py:139-140; logging_controller.py:158):
        # a missed detection is therefore logged as 0,0,0,0, not NaN.  Kept for log compatibility.
        boxes[~np.isfinite(boxes).all(axis=1)] = 0
        for i, box in enumerate(boxes):
            row = dict(frame=first + i, cycle=cycle, phase="imaging" if i < self.timing_config.imaging_frame_num else "moving")
            row["plt_x"], row["plt_y"] = plt[i]
            row["cam_x"], row["cam_y"], row["cam_w"], row["cam_h"] = cam[i]
            row["mic_x"], row["mic_y"], row["mic_w"], row["mic_h"] = mic[i]
            row["wrm_x"], row["wrm_y"], row["wrm_w"], row["wrm_h"] = box
            self.rows.append(row)
            if self._writer:
                self._writer.writerow(row)
        if self._file:
            self._file.flush()

    def on_sim_end(self, sim):
        self._flush_pending()
        self.sim_controller.on_sim_end(sim)
        if self._file:
            self._file.close()
            self._file = None

    def on_cycle_start(self, sim): self.sim_controller.on_cycle_start(sim)
    def on_imaging_start(self, sim): self.sim_controller.on_imaging_start(sim)
    def on_micro_frame(self, sim): self.sim_controller.on_micro_frame(sim)
    def on_imaging_end(self, sim): self.sim_controller.on_imaging_end(sim)
    def on_movement_start(self, sim): self.sim_controller.on_movement_start(sim)
    def on_movement_end(self, sim): self.sim_controller.on_movement_end(sim)
    def begin_movement_prediction(self, sim): return self.sim_controller.begin_movement_prediction(sim)
    def provide_movement_vector(self, sim): return self.sim_controller.provide_movement_vector(sim)
    def _cycle_predict_all(self, sim): return self.sim_controller._cycle_predict_all(sim)


def box_center(xywh: np.ndarray) -> np.ndarray:
    """(x + w / 2, y + h / 2) per row (BoxUtils.center, wtracker/utils/bbox_utils.py:76-92)."""
    b = np.asarray(xywh)
    return np.stack([b[..., 0] + b[..., 2] / 2, b[..., 1] + b[..., 3] / 2], axis=-1)


def xyxy_to_xywh(xyxy: np.ndarray) -> np.ndarray:
    """(x1, y1, x2 - x1, y2 - y1): what YoloController applies to the detector's box (BoxConverter.to_xywh from XYXY,
    bbox_utils.py:232-253; on the device this is head_select_kernel's epilogue)."""
    b = np.asarray(xyxy)
    return np.stack([b[..., 0], b[..., 1], b[..., 2] - b[..., 0], b[..., 3] - b[..., 1]], axis=-1)


def yolo_to_xywh(cxcywh: np.ndarray) -> np.ndarray:
    """Centre format -> corner format (BoxConverter.to_xywh from YOLO, bbox_utils.py:254-258)."""
    b = np.asarray(cxcywh)
    return np.stack([b[..., 0] - b[..., 2] / 2, b[..., 1] - b[..., 3] / 2, b[..., 2], b[..., 3]], axis=-1)


def discretize(bboxes: np.ndarray, bounds: tuple) -> tuple:
    """xywh float boxes -> int32 crop windows clamped to (H, W); illegal/NaN rows zeroed
    (BoxUtils.discretize, wtracker/utils/bbox_utils.py:118-167)."""
    b = np.array(bboxes, dtype=np.float64, copy=True)
    legal = np.isfinite(b).all(axis=1)
    b[~legal] = 0
    x1, y1 = np.floor(b[:, 0]), np.floor(b[:, 1])
    x2, y2 = np.ceil(b[:, 0] + b[:, 2]), np.ceil(b[:, 1] + b[:, 3])
    H, W = bounds
    x1, x2 = np.clip(x1, 0, W), np.clip(x2, 0, W)
    y1, y2 = np.clip(y1, 0, H), np.clip(y2, 0, H)
    w, h = x2 - x1, y2 - y1
    legal = (w > 0) & (h > 0)
    out = np.stack([x1, y1, w, h], axis=1)
    out[~legal] = 0
    return out.astype(np.int32), legal
