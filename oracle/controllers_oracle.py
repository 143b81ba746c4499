"""ORACLE (test infrastructure, never shipped, never imported by wtracker_amd/).

CPU restatements of the two hot-path controllers on top of the harness in wtracker_amd/sim.py:
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
from wtracker_amd.controllers import CsvController
from wtracker_amd.sim import SimController


class OracleMLPController(CsvController):
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
