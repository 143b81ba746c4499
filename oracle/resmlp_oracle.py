"""ORACLE (test infrastructure, never shipped, never imported by wtracker_amd/).

CPU restatement of the reference's ResMLP inference in numpy float32, layer by layer and WITHOUT
BatchNorm folding (so it is independent of the product's folding code):
  WormPredictor.forward -> RMLP.forward       wtracker/neural/mlp.py:47-48, 184-188
  MlpBlock.forward (4 MLPLayers, ReLU last)   wtracker/neural/mlp.py:121-126, 141
  MLPLayer = Linear -> BatchNorm1d(eval) -> ReLU   wtracker/neural/mlp.py:67-71, 89
and of MLPController.provide_movement_vector's arithmetic (mlp_controllers.py:36-68).

Pinned by tests/golden/resmlp_{100,200}ms.npz (weights, inputs and outputs captured from the real
reference by tests/golden/make_golden.py): see tests/test_oracle_resmlp.py.
"""
from __future__ import annotations

import re

import numpy as np

BN_EPS = np.float32(1e-5)


def _layer(sd, prefix, x):
    w = sd[prefix + ".0.weight"].astype(np.float32)
    b = sd[prefix + ".0.bias"].astype(np.float32)
    y = x @ w.T + b  # nn.Linear
    if prefix + ".1.running_mean" in sd:  # BatchNorm1d in eval mode: running statistics
        mu = sd[prefix + ".1.running_mean"].astype(np.float32)
        var = sd[prefix + ".1.running_var"].astype(np.float32)
        g = sd[prefix + ".1.weight"].astype(np.float32)
        beta = sd[prefix + ".1.bias"].astype(np.float32)
        y = (y - mu) / np.sqrt(var + BN_EPS) * g + beta
        y = np.maximum(y, np.float32(0))  # ReLU (both shipped models)
    return y.astype(np.float32)


def load_state(npz_path: str) -> dict:
    z = np.load(npz_path)
    sd = {k[4:]: z[k] for k in z.files if k.startswith("sd::")}
    return dict(sd=sd, input_frames=z["input_frames"].tolist(), pred_frames=z["pred_frames"].tolist())


def forward(state: dict, x: np.ndarray) -> np.ndarray:
    """x [B, 4*len(input_frames)] float32 -> [B, 2] float32."""
    sd = state["sd"]
    x = np.asarray(x, dtype=np.float32)
    h = _layer(sd, "model.input.mlp_layer", x)
    n_blocks = len({int(m.group(1)) for k in sd for m in [re.match(r"model\.blocks\.(\d+)\.", k)] if m})
    for b in range(n_blocks):
        t = h
        l = 0
        while f"model.blocks.{b}.sequence.{l}.mlp_layer.0.weight" in sd:
            t = _layer(sd, f"model.blocks.{b}.sequence.{l}.mlp_layer", t)
            l += 1
        h = h + t
    w = sd["model.output.weight"].astype(np.float32)
    bo = sd["model.output.bias"].astype(np.float32)
    return (h @ w.T + bo).astype(np.float32)


def movement_vector(state: dict, boxes7: np.ndarray, cam_position, max_dist_per_pred: float):
    """The arithmetic of MLPController.provide_movement_vector (mlp_controllers.py:41-65) given the 7
    gathered absolute boxes [7,4] float64 and the camera window (x, y, w, h)."""
    boxes = np.array(boxes7, dtype=np.float64).reshape(1, -1)
    if not np.isfinite(boxes).all():
        return 0, 0
    cam_cx, cam_cy = cam_position[0] + cam_position[2] / 2, cam_position[1] + cam_position[3] / 2
    x0, y0 = boxes[0, 0], boxes[0, 1]
    rel_x, rel_y = x0 - cam_cx, y0 - cam_cy
    boxes[:, 0::4] -= x0
    boxes[:, 1::4] -= y0
    pred = forward(state, boxes.astype(np.float32)).flatten()
    pred = np.clip(pred, -max_dist_per_pred, max_dist_per_pred)
    return round(pred[0].item() + rel_x), round(pred[1].item() + rel_y)


# ---- plain-C twin (oracle/resmlp_oracle.c), built by `make -C oracle` -----------------------------
def c_forward(state: dict, x: np.ndarray) -> np.ndarray:
    import ctypes as C
    import os

    lib_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libresmlp_oracle.so")
    lib = C.CDLL(lib_path)

    class Desc(C.Structure):
        _fields_ = [("in_dim", C.c_int), ("out_dim", C.c_int), ("has_bn", C.c_int)]

    sd = state["sd"]
    prefixes = ["model.input.mlp_layer"]
    n_blocks = len({int(m.group(1)) for k in sd for m in [re.match(r"model\.blocks\.(\d+)\.", k)] if m})
    per_block = 0
    while f"model.blocks.0.sequence.{per_block}.mlp_layer.0.weight" in sd:
        per_block += 1
    for b in range(n_blocks):
        for l in range(per_block):
            prefixes.append(f"model.blocks.{b}.sequence.{l}.mlp_layer")
    descs, blob = [], []
    for p in prefixes:
        w = sd[p + ".0.weight"].astype(np.float32)
        descs.append((w.shape[1], w.shape[0], 1))
        blob += [w.ravel(), sd[p + ".0.bias"], sd[p + ".1.weight"], sd[p + ".1.bias"], sd[p + ".1.running_mean"], sd[p + ".1.running_var"]]
    w = sd["model.output.weight"].astype(np.float32)
    descs.append((w.shape[1], w.shape[0], 0))
    blob += [w.ravel(), sd["model.output.bias"]]
    params = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.float32).ravel() for a in blob]))
    arr = (Desc * len(descs))(*[Desc(*d) for d in descs])
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty((x.shape[0], descs[-1][1]), dtype=np.float32)
    rc = lib.resmlp_forward(arr, n_blocks, per_block, params.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), x.shape[0],
                            y.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return y
