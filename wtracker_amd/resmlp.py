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


def _fold(prefix: str, sd: Mapping[str, np.ndarray]):
    """Linear at `prefix.0`, optional BatchNorm1d at `prefix.1` (present iff the layer has a nonlinearity,
    mlp.py:69-70) -> (W', b', relu) in float64 arithmetic, cast to float32."""
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
    # both shipped models use ReLU after every MLPLayer; a layer without BN has no
    # nonlinearity by construction (mlp.py:69)
    return w.astype(np.float32), b.astype(np.float32), bool(has_bn)


def fold_state_dict(sd: Mapping[str, np.ndarray], input_frames: Sequence[int], pred_frames: Sequence[int]) -> FoldedResMLP:
    sd = {k[4:] if k.startswith("sd::") else k: v for k, v in sd.items()}
    blocks = sorted({int(m.group(1)) for k in sd for m in [re.match(r"model\.blocks\.(\d+)\.", k)] if m})
    n_blocks = len(blocks)
    per_block = 0
    if n_blocks:
        per_block = len({int(m.group(1)) for k in sd for m in [re.match(r"model\.blocks\.0\.sequence\.(\d+)\.", k)] if m})
    layers = [_fold("model.input.mlp_layer", sd)]
    for b in range(n_blocks):
        for l in range(per_block):
            layers.append(_fold(f"model.blocks.{b}.sequence.{l}.mlp_layer", sd))
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
    return fold_state_dict(sd, list(model.io_config.input_frames), list(model.io_config.pred_frames))
