"""ORACLE (test infrastructure, never shipped, never imported by wtracker_amd/).

CPU restatements of the hot-path controllers.  They share NO arithmetic with the product's controllers (wtracker_amd/controllers.py);
only the hook-name ABC and the frame loop of the harness (wtracker_amd/sim.py) are common, and those are pinned by the real
reference's logs (tests/test_sim_golden.py) and by tests/golden/dropin_*.json (the product's controllers inside the REAL
reference's Simulator + LoggingController, tests/golden/make_golden.py --r3):
  OracleCsvController   <- CsvController   wtracker/sim/sim_controllers/csv_controller.py:11-73
  OracleMLPController   <- MLPController   wtracker/sim/sim_controllers/mlp_controllers.py:14-71
  OracleYoloController  <- YoloController  wtracker/sim/sim_controllers/yolo_controller.py:48-109
Pinned (MLP) by the golden logs tests/golden/sim_mlp*_bboxes.csv + sim_moves.json that the real
reference produced (tests/test_sim_golden.py).  The YOLO controller is parity-unpinned (see
oracle/yolo_oracle.py).
"""
from __future__ import annotations

from collections import deque

import numpy as np

from oracle import resmlp_oracle, view_oracle, yolo_oracle
from wtracker_amd.sim import SimController  # the hook-name ABC only (itself pinned by the reference's logs, tests/test_sim_golden.py)


class OracleCsvController(SimController):
    """Replay of a pre-detected track, restated for the checker WITHOUT the product's CsvController (wtracker_amd/controllers.py):
    plain per-frame Python over a list of rows and a list of camera corners, following csv_controller.py:11-73 step by step.
    The product's vectorised table / ring and this loop are two statements of the same gather; the golden logs of the real
    reference pin both (tests/test_sim_golden.py)."""

    def __init__(self, timing_config, csv_path: str):
        super().__init__(timing_config)
        import pandas as pd  # the reference's parser (csv_controller.py:16); see wtracker_amd.controllers._read_track_csv on why it matters

        table = pd.read_csv(csv_path, usecols=["wrm_x", "wrm_y", "wrm_w", "wrm_h"]).to_numpy(dtype=float)
        self.rows = [tuple(float(v) for v in r) for r in table]
        self.history: list = []  # camera corners of the frames seen so far, at most one cycle of them (oldest first)

    def on_sim_start(self, sim):
        self.history = []

    def on_camera_frame(self, sim):  # csv_controller.py:22-23
        self.history.append(tuple(float(v) for v in sim.view.camera_position))
        if len(self.history) > self.timing_config.cycle_frame_num:
            self.history.pop(0)

    def predict(self, frame_nums, relative: bool = True) -> np.ndarray:  # csv_controller.py:25-49
        assert len(frame_nums) > 0
        out = np.empty((len(frame_nums), 4), dtype=np.float64)
        for k, f in enumerate(int(v) for v in frame_nums):
            x, y, w, h = self.rows[f] if 0 <= f < len(self.rows) else (np.nan,) * 4
            if relative:
                cam = self.history[f % self.timing_config.cycle_frame_num]  # IndexError before the slot exists, like the reference
                x, y = x - cam[0], y - cam[1]
            out[k] = (x, y, w, h)
        return out

    def begin_movement_prediction(self, sim):
        pass

    def provide_movement_vector(self, sim):  # csv_controller.py:54-68
        x, y, w, h = self.predict([sim.frame_number - self.timing_config.pred_frame_num])[0]
        if not np.isfinite([x, y, w, h]).all():
            return 0, 0
        return round(x + w / 2 - sim.view.camera_size[0] / 2), round(y + h / 2 - sim.view.camera_size[1] / 2)

    def _cycle_predict_all(self, sim):  # csv_controller.py:69-73
        n = self.timing_config.cycle_frame_num
        first = (sim.cycle_number - 1) * n
        return self.predict(list(range(first, min(first + n, len(self.rows)))))


class OracleMLPController(OracleCsvController):
    def __init__(self, timing_config, csv_path: str, state: dict, max_speed: float = 0.9):
        super().__init__(timing_config, csv_path)
        self.state = state
        px_per_frame = max_speed * (timing_config.px_per_mm / timing_config.frames_per_sec)
        self.max_dist_per_pred = px_per_frame * state["pred_frames"][0]

    def provide_movement_vector(self, sim):
        frames = np.asanyarray(self.state["input_frames"], dtype=int) + (sim.frame_number - self.timing_config.pred_frame_num)
        boxes = self.predict(frames, relative=False)
        return resmlp_oracle.movement_vector(self.state, boxes, sim.view.camera_position, self.max_dist_per_pred)


class OracleYoloController(SimController):
    def __init__(self, timing_config, model: yolo_oracle.YoloOracle, imgsz: int = 384, conf: float = 0.1):
        super().__init__(timing_config)
        self._camera_frames = deque(maxlen=timing_config.cycle_frame_num)
        self._model, self.imgsz, self.conf = model, imgsz, conf

    def on_sim_start(self, sim):
        self._camera_frames.clear()

    def on_camera_frame(self, sim):
        # the view is cut by oracle/view_oracle.py from the raw frame, NOT by the product harness' ViewController:
        # an error in either statement of the slicing shows up as a difference in the closed-loop tests
        v = sim.view
        frame = v._frame_reader[v.index]
        self._camera_frames.append(view_oracle.camera_view(frame, tuple(int(p) for p in v.position), v.camera_size))

    def on_cycle_end(self, sim):
        self._camera_frames.clear()

    def predict(self, frames):
        return yolo_oracle.predict(self._model, list(frames), imgsz=self.imgsz, conf=self.conf)[0]

    def begin_movement_prediction(self, sim):
        pass

    def provide_movement_vector(self, sim):
        bbox = self.predict([self._camera_frames[-self.timing_config.pred_frame_num]])[0]
        if not np.isfinite(bbox).all():
            return 0, 0
        mid = bbox[0] + bbox[2] / 2, bbox[1] + bbox[3] / 2
        return round(mid[0] - sim.view.camera_size[0] / 2), round(mid[1] - sim.view.camera_size[1] / 2)

    def _cycle_predict_all(self, sim):
        return self.predict(self._camera_frames)
