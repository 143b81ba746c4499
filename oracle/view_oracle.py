"""ORACLE (test infrastructure, never shipped, never imported by wtracker_amd/).

numpy restatement of the camera / microscope view extraction that feeds the detector in the reference's loop:

  ViewController.read            wtracker/sim/view_controller.py:45-61   frame padded by camera_size // 2 on every side,
                                                                         cv.copyMakeBorder(BORDER_REPLICATE) == np.pad(mode="edge")
  camera_position / micro_position   :93-117                             (x - w // 2, y - h // 2, w, h) in UNPADDED coordinates
  set_position                   :119-131                                position clamped to [0, W-1] x [0, H-1] of the unpadded frame
  _calc_view_bbox                :143-156                                x = pos_x + pad_x - w // 2, y = pos_y + pad_y - h // 2
  _custom_view                   :158-172                                frame[y : y + w, x : x + h]   (rows by w, columns by h)

Written independently of wtracker_amd.sim.ViewController (the product harness) so that the closed-loop tests compare two
separate statements of the slicing.  cv2 is absent here; BORDER_REPLICATE has a single possible meaning (edge value
repeated), so this part is pinned by definition, unlike the bilinear resize of the letterbox (yolo_oracle.resize_bilinear_u8).
"""
from __future__ import annotations

import numpy as np


def clamp_position(x: int, y: int, frame_hw: tuple) -> tuple:
    """set_position (view_controller.py:119-131)."""
    return int(np.clip(x, 0, frame_hw[1] - 1)), int(np.clip(y, 0, frame_hw[0] - 1))


def padded(frame: np.ndarray, camera_size: tuple) -> np.ndarray:
    """read (view_controller.py:45-61): left/right = camera_size[0] // 2, top/bottom = camera_size[1] // 2, replicate."""
    px, py = camera_size[0] // 2, camera_size[1] // 2
    pad = ((py, py), (px, px)) + (((0, 0),) if frame.ndim == 3 else ())
    return np.pad(frame, pad, mode="edge")


def custom_view(frame: np.ndarray, position: tuple, camera_size: tuple, size: tuple) -> np.ndarray:
    """_calc_view_bbox + _custom_view (view_controller.py:143-172) for a view of `size` = (w, h)."""
    w, h = size
    x = position[0] + camera_size[0] // 2 - w // 2
    y = position[1] + camera_size[1] // 2 - h // 2
    return padded(frame, camera_size)[y : y + w, x : x + h]


def camera_view(frame: np.ndarray, position: tuple, camera_size: tuple) -> np.ndarray:
    return custom_view(frame, position, camera_size, camera_size)


def micro_view(frame: np.ndarray, position: tuple, camera_size: tuple, micro_size: tuple) -> np.ndarray:
    return custom_view(frame, position, camera_size, micro_size)
