"""ResMLP parameter handling: fold BatchNorm1d (eval mode) into the preceding Linear.

The reference's predictor is a whole-module pickle of `WormPredictor(RMLP)`
(wtracker/neural/mlp.py:31-48,144-188; loaded at workflows/simulate.ipynb cell 9).  This module
works from its *state dict* only (numeric data), so it never needs the reference's classes:
keys `model.input.mlp_layer.{0,1}.*`, `model.blocks.{b}.sequence.{l}.mlp_layer.{0,1}.*`,
`model.output.*`.
"""
from __future__ import annotations

import re
from dataclasses import dataclass
from typing import Mapping, Sequence

import numpy as np

from .hip import WtkError

BN_EPS = 1e-5  # torch.nn.BatchNorm1d default, used by MLPLayer (mlp.py:70)


@dataclass
class FoldedResMLP:
    """Affine layers in execution order: input, block0.l0.., ..., output; each (W[out,in], b[out], relu)."""

    layers: list
    n_blocks: int
    layers_per_block: int
    input_frames: list
    pred_frames: list

    @property
    def in_dim(self) -> int:
        return int(self.layers[0][0].shape[1])

    @property
    def out_dim(self) -> int:
        return int(self.layers[-1][0].shape[0])

    @property
    def macs_per_sample(self) -> int:
        return int(sum(w.shape[0] * w.shape[1] for w, _, _ in self.layers))


def _fold(prefix: str, sd: Mapping[str, np.ndarray], activation: str | None = None):
    """Linear at `prefix.0`, optional BatchNorm1d at `prefix.1` -> (W', b', relu) in float64 arithmetic, cast to
    float32.  `activation` ('relu' / 'none') is the layer's nonlinearity when the caller knows it (from_torch_module
    reads it off the live module); None = infer it from the state dict: MLPLayer adds a BatchNorm1d exactly when it has a
    nonlinearity and batch_norm=True (mlp.py:69-70), which is how both shipped models and the fixtures are built."""
    w = np.asarray(sd[prefix + ".0.weight"], dtype=np.float64)
    b = np.asarray(sd[prefix + ".0.bias"], dtype=np.float64)
    has_bn = (prefix + ".1.running_mean") in sd
    if has_bn:
        g = np.asarray(sd[prefix + ".1.weight"], dtype=np.float64)
        beta = np.asarray(sd[prefix + ".1.bias"], dtype=np.float64)
        mu = np.asarray(sd[prefix + ".1.running_mean"], dtype=np.float64)
        var = np.asarray(sd[prefix + ".1.running_var"], dtype=np.float64)
        s = g / np.sqrt(var + BN_EPS)
        w = w * s[:, None]
        b = (b - mu) * s + beta
    if activation is None:
        relu = bool(has_bn)  # state dict only: BN present <=> ReLU layer of the shipped models
    elif activation in ("relu", "none"):
        relu = activation == "relu"
    else:
        raise WtkError(f"ResMLP layer {prefix}: activation {activation!r} is not supported by mlp_kernel (ReLU or none only)")
    return w.astype(np.float32), b.astype(np.float32), relu


def fold_state_dict(sd: Mapping[str, np.ndarray], input_frames: Sequence[int], pred_frames: Sequence[int],
                    activations: Sequence[str] | None = None) -> FoldedResMLP:
    """`activations`: one entry ('relu' / 'none') per MLPLayer in execution order (input, block0.l0, ...); None infers
    ReLU from the presence of a BatchNorm (see _fold)."""
    sd = {k[4:] if k.startswith("sd::") else k: v for k, v in sd.items()}
    blocks = sorted({int(m.group(1)) for k in sd for m in [re.match(r"model\.blocks\.(\d+)\.", k)] if m})
    n_blocks = len(blocks)
    per_block = 0
    if n_blocks:
        per_block = len({int(m.group(1)) for k in sd for m in [re.match(r"model\.blocks\.0\.sequence\.(\d+)\.", k)] if m})
    prefixes = ["model.input.mlp_layer"] + [f"model.blocks.{b}.sequence.{l}.mlp_layer" for b in range(n_blocks) for l in range(per_block)]
    if activations is not None and len(activations) != len(prefixes):
        raise WtkError(f"ResMLP: {len(activations)} activations given for {len(prefixes)} layers")
    layers = [_fold(p, sd, None if activations is None else activations[i]) for i, p in enumerate(prefixes)]
    w = np.asarray(sd["model.output.weight"], dtype=np.float32)
    bo = np.asarray(sd["model.output.bias"], dtype=np.float32)
    layers.append((w, bo, False))
    return FoldedResMLP(layers, n_blocks, per_block, [int(v) for v in input_frames], [int(v) for v in pred_frames])


def load_npz(path: str) -> FoldedResMLP:
    """Load a fixture written by tests/golden/make_golden.py (state-dict tensors + IOConfig)."""
    z = np.load(path)
    sd = {k: z[k] for k in z.files if k.startswith("sd::")}
    return fold_state_dict(sd, z["input_frames"].tolist(), z["pred_frames"].tolist())


def from_torch_module(model) -> FoldedResMLP:
    """From a live `WormPredictor` (what simulate.ipynb cell 9 passes to MLPController): uses only
    `state_dict()` and `io_config.{input_frames,pred_frames}`."""
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    # the nonlinearity is not in the state dict: read it off every MLPLayer (`mlp_layer` = Sequential(Linear, [BN], act),
    # mlp.py:67-73).  Anything but ReLU / Identity would be computed wrongly by the device kernel -> refuse it.
    acts = []
    for name, mod in model.named_modules():
        seq = getattr(mod, "mlp_layer", None)
        if seq is None:
            continue
        kind = type(list(seq.children())[-1]).__name__
        if kind not in ("ReLU", "Identity"):
            raise WtkError(f"ResMLP layer {name}: activation {kind} is not supported by mlp_kernel (ReLU or none only)")
        acts.append("relu" if kind == "ReLU" else "none")
    return fold_state_dict(sd, list(model.io_config.input_frames), list(model.io_config.pred_frames), acts or None)


def make_training_pairs(log_path: str, input_frames: Sequence[int], pred_frames: Sequence[int]):
    """(X [n, 4*len(input_frames)], y [n, 2*len(pred_frames)]) float32 training pairs from a bboxes.csv
    log — the arithmetic of NumpyDataset.create_from_config (wtracker/neural/dataset.py:42-96), vectorised
    (the reference fills two object DataFrames row by row).

    Row i (abs(min(input_frames)) + 1 <= i < len - max(pred_frames) - 1) holds the xywh boxes at
    i + input_frames and the box centres at i + pred_frames; rows with any NaN are dropped; values are
    cast to float32 FIRST and only then made relative to the frame-i box corner (that order is part of
    the result's bits)."""
    import pandas as pd

    data = pd.read_csv(log_path)
    boxes = data[["wrm_x", "wrm_y", "wrm_w", "wrm_h"]].to_numpy(dtype=np.float64)
    centers = np.stack([boxes[:, 0] + boxes[:, 2] / 2, boxes[:, 1] + boxes[:, 3] / 2], axis=1)
    xi = np.asanyarray(input_frames, dtype=int)
    yi = np.asanyarray(pred_frames, dtype=int)
    rows = np.arange(abs(int(xi.min())) + 1, len(boxes) - int(yi.max()) - 1)
    if rows.size == 0:
        return np.empty((0, 4 * len(xi)), np.float32), np.empty((0, 2 * len(yi)), np.float32)
    X = boxes[rows[:, None] + xi[None, :]].reshape(rows.size, -1)
    y = centers[rows[:, None] + yi[None, :]].reshape(rows.size, -1)
    keep = ~(np.isnan(X).any(axis=1) | np.isnan(y).any(axis=1))
    X = X[keep].astype(np.float32)
    y = y[keep].astype(np.float32)
    x0 = X[:, 0:1].copy()
    y0 = X[:, 1:2].copy()
    y[:, 0::2] -= x0
    y[:, 1::2] -= y0
    X[:, 0::4] -= x0
    X[:, 1::4] -= y0
    return X, y


def make_training_pairs_device(track_dev, input_frames: Sequence[int], pred_frames: Sequence[int]):
    """make_training_pairs on a DEVICE-resident track (torch CUDA tensor [N,4], float64 as loaded from a bboxes.csv or the
    detector's float32 track): wtk_track_training_pairs builds every candidate row, the NaN rows are dropped on the device.
    Returns (X, y) as float32 CUDA tensors — bit-identical to the reference's NumpyDataset for a float64 track."""
    import torch

    from . import hip

    n = int(track_dev.shape[0])
    xi, yi = [int(v) for v in input_frames], [int(v) for v in pred_frames]
    row0 = abs(min(xi)) + 1
    n_rows = max(n - max(yi) - 1 - row0, 0)
    dev = track_dev.device
    X = torch.empty((n_rows, 4 * len(xi)), dtype=torch.float32, device=dev)
    y = torch.empty((n_rows, 2 * len(yi)), dtype=torch.float32, device=dev)
    if n_rows == 0:
        return X, y
    keep = torch.empty((n_rows,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        hip.track_training_pairs(track_dev.contiguous(), n, row0, n_rows, xi, yi, X, y, keep, stream=torch.cuda.current_stream(dev).cuda_stream)
    m = keep.bool()
    return X[m], y[m]
