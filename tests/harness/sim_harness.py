"""Stand-in DRIVER for `SimController` plugins: test infrastructure, not product.

The product (wtracker_amd) implements controllers behind the reference's plugin interface; the thing that CALLS the hooks is the
reference's `Simulator` (wtracker/sim/simulator.py:140-194).  The reference cannot travel to the GPU box, so the GPU tests and
bench.py's closed-loop leg drive the controllers with this harness instead.  It is pinned, not trusted: tests/test_sim_golden.py
requires the logs it produces to equal the logs the real reference wrote (tests/golden/sim_*_bboxes.csv, sim_moves.json, every row
and every integer move), and tests/test_dropin_reference.py requires its log file's sha256 to equal the reference driver's.

What it states (behaviour, each pinned by those fixtures):
  * hook order per frame of a cycle            wtracker/sim/simulator.py:157-184   -> `_hook_plan` (a table, built once)
  * platform motion: half-cosine profile, each step rounded, the rounding residual carried into the next step
                                               wtracker/sim/motor_controllers.py:58-88 -> `SineMotorController` (a cursor + carry)
  * camera / microscope windows on a replicate-padded frame, position clamped to the unpadded frame
                                               wtracker/sim/view_controller.py:45-172  -> `ViewController`
  * DummyReader's blank frames                 wtracker/sim/frame_reader.py:247-272   -> `BlankReader`
"""
from __future__ import annotations

import numpy as np

from wtracker_amd.sim import ExperimentConfig, SimController, TimingConfig


class BlankReader:
    """Length provider with all-255 frames."""

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
        # clamped to the UNPADDED frame extent
        x = np.clip(x, 0, self._frame_reader.frame_shape[1] - 1)
        y = np.clip(y, 0, self._frame_reader.frame_shape[0] - 1)
        self._position = (x, y)

    def move_position(self, dx, dy):
        self.set_position(self._position[0] + dx, self._position[1] + dy)

    def read(self) -> np.ndarray:
        """Current frame with a replicate border of camera_size // 2."""
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
        # the reference slices rows by w and columns by h (view_controller.py:171); the golden logs were made that way
        return self.read()[y : y + w, x : x + h]

    def camera_view(self) -> np.ndarray:
        return self._view(self._camera_size)

    def micro_view(self) -> np.ndarray:
        return self._view(self._micro_size)


class SineMotorController:
    """Platform motion over `moving_frame_num` frames along a half-cosine velocity profile.  Step k covers the fraction
    (cos(k pi / n) - cos((k + 1) pi / n)) / 2 of the move; the platform takes whole pixels, and what rounding leaves over is carried
    into the next step, so the steps of a move always sum to the requested vector.  State = (the move, a cursor, the carry)."""

    def __init__(self, timing_config: TimingConfig):
        self.timing_config = timing_config
        self.movement_steps = n = timing_config.moving_frame_num
        # the profile depends on n only (np.cos, as the fixtures' generator used)
        self._share = [(np.cos((k * np.pi) / n) - np.cos(((k + 1) * np.pi) / n)) / 2 for k in range(n)]
        self._move = (0, 0)
        self._cursor = n  # == n: no move in progress
        self._carry = (0.0, 0.0)

    @property
    def busy(self) -> bool:
        return self._cursor < self.movement_steps

    def register_move(self, dx, dy):
        assert not self.busy, "the previous move has not finished"
        self._move, self._cursor, self._carry = (dx, dy), 0, (0.0, 0.0)

    def step(self):
        share = self._share[self._cursor]
        self._cursor += 1
        want = (share * self._move[0] + self._carry[0], share * self._move[1] + self._carry[1])
        took = (round(want[0]), round(want[1]))
        self._carry = (want[0] - took[0], want[1] - took[1])
        return took


def _hook_plan(tc: TimingConfig) -> list:
    """Per step of a cycle: (hooks fired for the frame, in order; is this the step the movement vector is asked for; does the platform move).
    Cycle boundaries (movement_end / cycle_end of the previous cycle, cycle_start) are handled by the loop: they depend on the cycle number."""
    img, mov, pred = tc.imaging_frame_num, tc.moving_frame_num, tc.pred_frame_num
    plan = []
    for step in range(tc.cycle_frame_num):
        hooks = ["on_camera_frame"]
        if step == 0:
            hooks.append("on_imaging_start")
        if step < img:
            hooks.append("on_micro_frame")
        if step == img - pred:
            hooks.append("begin_movement_prediction")
        plan.append((tuple(hooks), step == img, img <= step < img + mov))
    return plan


class Simulator:
    """Drives a SimController over the frames of a reader: per frame the hooks of `_hook_plan`, at the end of the imaging phase the
    controller's movement vector goes to the motor, during the moving phase the motor's steps move the view."""

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
        ctl, motor, view = self._sim_controller, self._motor_controller, self._view
        plan = _hook_plan(self.timing_config)
        cyc = self.timing_config.cycle_frame_num
        view.reset()
        view.set_position(*self.experiment_config.init_position)
        ctl.on_sim_start(self)
        n_frames = len(view)
        for frame in range(n_frames):
            view.seek(frame)
            cycle, step = divmod(frame, cyc)
            if step == 0:
                if cycle > 0:  # the previous cycle ends when the next one's first frame arrives (the last cycle is never closed)
                    ctl.on_movement_end(self)
                    ctl.on_cycle_end(self)
                ctl.on_cycle_start(self)
            hooks, decide, moving = plan[step]
            for name in hooks:
                getattr(ctl, name)(self)
            if decide:
                ctl.on_imaging_end(self)
                vector = ctl.provide_movement_vector(self)
                ctl.on_movement_start(self)
                motor.register_move(*vector)
            if moving:
                view.move_position(*motor.step())
        view.seek(n_frames)  # one past the end, where the reference's cursor stops
        ctl.on_sim_end(self)
