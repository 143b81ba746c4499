"""Minimal host-side harness that drives a controller exactly the way the reference's simulator
drives a `SimController`, so the HIP controllers can be exercised (and compared with golden logs
captured from the real reference) without the reference being present.

It restates ONLY what the hot path's callers need:
  * TimingConfig / ExperimentConfig arithmetic           wtracker/sim/config.py:41-67, 105-111
  * the frame loop + phase state machine + hook order    wtracker/sim/simulator.py:140-194
  * the SimController plugin interface                   wtracker/sim/simulator.py:197-293
  * platform motion profile (half cosine, residual carry) wtracker/sim/motor_controllers.py:58-88
  * camera / microscope windows on a replicate-padded frame  wtracker/sim/view_controller.py:45-172
  * the per-cycle track log (CSV schema)                 wtracker/sim/sim_controllers/logging_controller.py:95-116,145-185
Image saving, tqdm, GUI prompts, disk readers are out of scope.
"""
from __future__ import annotations

import abc
import csv
import math
from collections import deque
from dataclasses import dataclass
from typing import Optional, Sequence

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
# frame sources
# -------------------------------------------------------------------------------------------------
class BlankReader:
    """Length provider with all-255 frames (the reference's DummyReader, frame_reader.py:247-272)."""

    def __init__(self, num_frames: int, resolution: tuple, colored: bool = True):
        self._n = num_frames
        self.frame_shape = (*resolution, 3) if colored else tuple(resolution)
        self._frame = np.full(self.frame_shape, 255, dtype=np.uint8)

    def __len__(self):
        return self._n

    def __getitem__(self, idx: int) -> np.ndarray:
        return self._frame.copy()


class ArrayReader:
    """Frames held in memory: uint8 [N,H,W] or [N,H,W,3]."""

    def __init__(self, frames: np.ndarray):
        assert frames.dtype == np.uint8 and frames.ndim in (3, 4)
        self._frames = frames
        self.frame_shape = tuple(frames.shape[1:])

    def __len__(self):
        return self._frames.shape[0]

    def __getitem__(self, idx: int) -> np.ndarray:
        if idx < 0 or idx >= len(self):
            raise IndexError("index out of bounds")
        return self._frames[idx]


class ViewController:
    """Cursor over a frame source + camera / microscope windows centred on the platform position."""

    def __init__(self, frame_reader, camera_size=(251, 251), micro_size=(45, 45), init_position=(0, 0)):
        assert camera_size[0] >= micro_size[0] and camera_size[1] >= micro_size[1]
        self._frame_reader = frame_reader
        self._idx = -1
        self._camera_size = tuple(camera_size)
        self._micro_size = tuple(micro_size)
        self._pad = (camera_size[0] // 2, camera_size[1] // 2)
        self._position = tuple(init_position)
        self.set_position(*init_position)

    # cursor
    @property
    def index(self) -> int:
        return self._idx

    def __len__(self):
        return len(self._frame_reader)

    def can_read(self) -> bool:
        return 0 <= self._idx < len(self._frame_reader)

    def seek(self, idx: int) -> bool:
        self._idx = idx
        return self.can_read()

    def progress(self, n: int = 1) -> bool:
        return self.seek(self._idx + n)

    def reset(self):
        self.seek(-1)

    # geometry
    @property
    def position(self):
        return self._position

    @property
    def camera_size(self):
        return self._camera_size

    @property
    def micro_size(self):
        return self._micro_size

    def _window(self, size):
        w, h = size
        return self._position[0] - w // 2, self._position[1] - h // 2, w, h

    @property
    def camera_position(self):
        return self._window(self._camera_size)

    @property
    def micro_position(self):
        return self._window(self._micro_size)

    def set_position(self, x, y):
        # clamped to the UNPADDED frame extent (view_controller.py:129-131)
        x = np.clip(x, 0, self._frame_reader.frame_shape[1] - 1)
        y = np.clip(y, 0, self._frame_reader.frame_shape[0] - 1)
        self._position = (x, y)

    def move_position(self, dx, dy):
        self.set_position(self._position[0] + dx, self._position[1] + dy)

    def read(self) -> np.ndarray:
        """Current frame with a replicate border of camera_size//2 (cv.copyMakeBorder BORDER_REPLICATE)."""
        if not self.can_read():
            raise IndexError("index out of bounds")
        f = self._frame_reader[self._idx]
        px, py = self._pad
        pad = ((py, py), (px, px)) + (((0, 0),) if f.ndim == 3 else ())
        return np.pad(f, pad, mode="edge")

    def _view(self, size) -> np.ndarray:
        w, h = size
        x = self._position[0] + self._pad[0] - w // 2
        y = self._position[1] + self._pad[1] - h // 2
        # the reference slices rows by w and columns by h (view_controller.py:171); kept as is
        return self.read()[y : y + w, x : x + h]

    def camera_view(self) -> np.ndarray:
        return self._view(self._camera_size)

    def micro_view(self) -> np.ndarray:
        return self._view(self._micro_size)


# -------------------------------------------------------------------------------------------------
# platform motor
# -------------------------------------------------------------------------------------------------
class SineMotorController:
    """Half-cosine velocity profile; each step is rounded and its residual carried to the next."""

    def __init__(self, timing_config: TimingConfig):
        self.timing_config = timing_config
        self.movement_steps = timing_config.moving_frame_num
        self.queue: list = []

    def register_move(self, dx, dy):
        assert len(self.queue) == 0
        n = self.movement_steps
        for i in range(n):
            frac = (np.cos((i * np.pi) / n) - np.cos(((i + 1) * np.pi) / n)) / 2
            self.queue.append((frac * dx, frac * dy))

    def step(self):
        dx, dy = self.queue.pop(0)
        rdx, rdy = round(dx), round(dy)
        if self.queue:
            nx, ny = self.queue[0]
            self.queue[0] = (nx + (dx - rdx), ny + (dy - rdy))
        return rdx, rdy


# -------------------------------------------------------------------------------------------------
# plugin interface + driver
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


class Simulator:
    def __init__(self, timing_config: TimingConfig, experiment_config: ExperimentConfig, sim_controller: SimController,
                 reader=None, motor_controller=None):
        self.timing_config = timing_config
        self.experiment_config = experiment_config
        self._sim_controller = sim_controller
        if reader is None:
            cam = timing_config.camera_size_px
            pad = (cam[0] // 2 * 2, cam[1] // 2 * 2)
            res = tuple(a + b for a, b in zip(experiment_config.orig_resolution, pad))
            reader = BlankReader(experiment_config.num_frames, res, colored=True)
        self._motor_controller = motor_controller or SineMotorController(timing_config)
        self._view = ViewController(reader, timing_config.camera_size_px, timing_config.micro_size_px, experiment_config.init_position)

    @property
    def view(self) -> ViewController:
        return self._view

    @property
    def position(self):
        return self._view.position

    @property
    def frame_number(self) -> int:
        return self._view.index

    @property
    def cycle_number(self) -> int:
        return self._view.index // self.timing_config.cycle_frame_num

    @property
    def cycle_step(self) -> int:
        return self._view.index % self.timing_config.cycle_frame_num

    def camera_view(self) -> np.ndarray:
        return self._view.camera_view()

    def micro_view(self) -> np.ndarray:
        return self._view.micro_view()

    def run(self, visualize: bool = False, wait_key: bool = False):
        tc, ctl, motor = self.timing_config, self._sim_controller, self._motor_controller
        self._view.reset()
        self._view.set_position(*self.experiment_config.init_position)
        ctl.on_sim_start(self)
        while self._view.progress():
            step = self.cycle_step
            if step == 0:
                if self.cycle_number > 0:
                    ctl.on_movement_end(self)
                    ctl.on_cycle_end(self)
                ctl.on_cycle_start(self)
            ctl.on_camera_frame(self)
            if step == 0:
                ctl.on_imaging_start(self)
            if step < tc.imaging_frame_num:
                ctl.on_micro_frame(self)
            if step == tc.imaging_frame_num - tc.pred_frame_num:
                ctl.begin_movement_prediction(self)
            if step == tc.imaging_frame_num:
                ctl.on_imaging_end(self)
                dx, dy = ctl.provide_movement_vector(self)
                ctl.on_movement_start(self)
                motor.register_move(dx, dy)
            if tc.imaging_frame_num <= step < tc.imaging_frame_num + tc.moving_frame_num:
                dx, dy = motor.step()
                self._view.move_position(dx, dy)
        ctl.on_sim_end(self)


# -------------------------------------------------------------------------------------------------
# track log (the on-disk wire format of tracks: bboxes.csv / init_bboxes.csv)
# -------------------------------------------------------------------------------------------------
LOG_COLUMNS = ["frame", "cycle", "phase", "plt_x", "plt_y", "cam_x", "cam_y", "cam_w", "cam_h",
               "mic_x", "mic_y", "mic_w", "mic_h", "wrm_x", "wrm_y", "wrm_w", "wrm_h"]


class TrackLogger(SimController):
    """Decorator controller: forwards every hook and, at each cycle end, asks the wrapped controller
    for the whole cycle's boxes (`_cycle_predict_all` -> the batched detector call), makes them
    absolute and appends one CSV row per frame.  Rows are also kept in `self.rows`."""

    def __init__(self, sim_controller: SimController, csv_path: Optional[str] = None):
        super().__init__(sim_controller.timing_config)
        self.sim_controller = sim_controller
        self.csv_path = csv_path
        n = self.timing_config.cycle_frame_num
        self._plt, self._cam, self._mic = deque(maxlen=n), deque(maxlen=n), deque(maxlen=n)
        self.rows: list = []
        self._file = None
        self._writer = None

    def on_sim_start(self, sim):
        self.sim_controller.on_sim_start(sim)
        self._plt.clear(), self._cam.clear(), self._mic.clear()
        self.rows = []
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
        first = cycle * self.timing_config.cycle_frame_num
        boxes = self.sim_controller._cycle_predict_all(sim)
        cams = np.asanyarray(list(self._cam))
        boxes[:, 0] += cams[:, 0]
        boxes[:, 1] += cams[:, 1]
        # The reference calls BoxUtils.discretize on this array before writing the rows, and that
        # function zeroes non-finite rows IN PLACE (bbox_utils.py:139-140; logging_controller.py:158):
        # a missed detection is therefore logged as 0,0,0,0, not NaN.  Kept for log compatibility.
        boxes[~np.isfinite(boxes).all(axis=1)] = 0
        for i, box in enumerate(boxes):
            row = dict(frame=first + i, cycle=cycle, phase="imaging" if i < self.timing_config.imaging_frame_num else "moving")
            row["plt_x"], row["plt_y"] = self._plt[i]
            row["cam_x"], row["cam_y"], row["cam_w"], row["cam_h"] = self._cam[i]
            row["mic_x"], row["mic_y"], row["mic_w"], row["mic_h"] = self._mic[i]
            row["wrm_x"], row["wrm_y"], row["wrm_w"], row["wrm_h"] = box
            self.rows.append(row)
            if self._writer:
                self._writer.writerow(row)
        if self._file:
            self._file.flush()
        self.sim_controller.on_cycle_end(sim)
        self._plt.clear(), self._cam.clear(), self._mic.clear()

    def on_sim_end(self, sim):
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
