"""YOLOv8 detector description: fused-conv table, synthetic seeded weights, weight-file I/O.

The conv table restates the public YOLOv8 architecture at scale (width, depth, max_channels) with
`nc` classes (ultralytics `yolov8.yaml`, third-party, not vendored by the reference).
It must equal the table libwtk_hip.so builds for itself (tests/test_abi.py checks that).

Weight convention everywhere in this repo: O-H-W-I = [cout][kh][kw][cin] fp32, BatchNorm folded,
stem input channels in RGB order.
"""
from __future__ import annotations

import json
import math
import os
import struct
from typing import Dict, Tuple

import numpy as np

SCALES = {
    # name: (depth, width, max_channels)
    "n": (0.33, 0.25, 1024),
    "s": (0.33, 0.50, 1024),
    "m": (0.67, 0.75, 768),
}
REG_MAX = 16
STRIDES = (8, 16, 32)


def make_divisible8(x: float) -> int:
    return int(math.ceil(x / 8.0) * 8)


def model_dims(width: float, depth: float, max_channels: int, nc: int) -> dict:
    base = (64, 128, 256, 512, 1024)
    c = [make_divisible8(min(b, max_channels) * width) for b in base]
    n = [max(round(r * depth), 1) for r in (3, 6, 6, 3)]
    hb = max(16, c[2] // 4, REG_MAX * 4)
    hc = max(c[2], min(nc, 100))
    return dict(c=c, n=n, hb=hb, hc=hc, nc=nc)


def _c2f(v, p, c1, c2, n):
    c = c2 // 2
    v.append(dict(name=p + ".cv1", cout=2 * c, cin=c1, k=1, stride=1, act=1))
    v.append(dict(name=p + ".cv2", cout=c2, cin=(2 + n) * c, k=1, stride=1, act=1))
    for i in range(n):
        v.append(dict(name=f"{p}.m.{i}.cv1", cout=c, cin=c, k=3, stride=1, act=1))
        v.append(dict(name=f"{p}.m.{i}.cv2", cout=c, cin=c, k=3, stride=1, act=1))


def conv_table(scale: str | tuple = "s", nc: int = 1) -> list[dict]:
    depth, width, maxch = SCALES[scale] if isinstance(scale, str) else scale
    d = model_dims(width, depth, maxch, nc)
    c, n = d["c"], d["n"]
    v: list[dict] = []
    v.append(dict(name="model.0", cout=c[0], cin=3, k=3, stride=2, act=1))
    v.append(dict(name="model.1", cout=c[1], cin=c[0], k=3, stride=2, act=1))
    _c2f(v, "model.2", c[1], c[1], n[0])
    v.append(dict(name="model.3", cout=c[2], cin=c[1], k=3, stride=2, act=1))
    _c2f(v, "model.4", c[2], c[2], n[1])
    v.append(dict(name="model.5", cout=c[3], cin=c[2], k=3, stride=2, act=1))
    _c2f(v, "model.6", c[3], c[3], n[2])
    v.append(dict(name="model.7", cout=c[4], cin=c[3], k=3, stride=2, act=1))
    _c2f(v, "model.8", c[4], c[4], n[3])
    v.append(dict(name="model.9.cv1", cout=c[4] // 2, cin=c[4], k=1, stride=1, act=1))
    v.append(dict(name="model.9.cv2", cout=c[4], cin=c[4] * 2, k=1, stride=1, act=1))
    _c2f(v, "model.12", c[4] + c[3], c[3], n[3])
    _c2f(v, "model.15", c[3] + c[2], c[2], n[3])
    v.append(dict(name="model.16", cout=c[2], cin=c[2], k=3, stride=2, act=1))
    _c2f(v, "model.18", c[2] + c[3], c[3], n[3])
    v.append(dict(name="model.19", cout=c[3], cin=c[3], k=3, stride=2, act=1))
    _c2f(v, "model.21", c[3] + c[4], c[4], n[3])
    ch = (c[2], c[3], c[4])
    for i in range(3):
        p = f"model.22.cv2.{i}"
        v.append(dict(name=p + ".0", cout=d["hb"], cin=ch[i], k=3, stride=1, act=1))
        v.append(dict(name=p + ".1", cout=d["hb"], cin=d["hb"], k=3, stride=1, act=1))
        v.append(dict(name=p + ".2", cout=4 * REG_MAX, cin=d["hb"], k=1, stride=1, act=0))
    for i in range(3):
        p = f"model.22.cv3.{i}"
        v.append(dict(name=p + ".0", cout=d["hc"], cin=ch[i], k=3, stride=1, act=1))
        v.append(dict(name=p + ".1", cout=d["hc"], cin=d["hc"], k=3, stride=1, act=1))
        v.append(dict(name=p + ".2", cout=nc, cin=d["hc"], k=1, stride=1, act=0))
    return v


def scale_params(scale: str | tuple) -> tuple[float, float, int]:
    """-> (width, depth, max_channels) in the order the C ABI takes them."""
    depth, width, maxch = SCALES[scale] if isinstance(scale, str) else scale
    return width, depth, maxch


def split_capable(scale: str | tuple, nc: int = 1) -> bool:
    """True when the detector's "f16x3" mode (split-fp16 operands) exists for this scale: every channel width a multiple of 64
    (stem 32), head widths multiples of 32 — YOLOv8 s and l (wtk_yolo_create checks the same)."""
    depth, width, maxch = SCALES[scale] if isinstance(scale, str) else scale
    d = model_dims(width, depth, maxch, nc)
    c = d["c"]
    return all(v % 64 == 0 or (i == 0 and v == 32) for i, v in enumerate(c)) and d["hb"] % 32 == 0 and d["hc"] % 32 == 0


def macs_per_frame(scale: str | tuple, nc: int, H: int, W: int) -> float:
    """Conv multiply-accumulates of one forward pass (2 FLOP each); SURVEY.md §8d: 14.216 G at s/nc=1/640^2."""
    total = 0.0
    depth, width, maxch = SCALES[scale] if isinstance(scale, str) else scale
    d = model_dims(width, depth, maxch, nc)
    c = d["c"]
    res = {}  # output stride of each conv
    for t in conv_table(scale, nc):
        nm = t["name"]
        idx = int(nm.split(".")[1])
        if idx == 0:
            s = 2
        elif idx in (1, 2):
            s = 4
        elif idx in (3, 4, 15):
            s = 8
        elif idx in (5, 6, 12, 16, 18):
            s = 16
        elif idx in (7, 8, 9, 19, 21):
            s = 32
        elif idx == 22:
            s = STRIDES[int(nm.split(".")[3])]
        else:
            raise AssertionError(nm)
        total += (H // s) * (W // s) * t["cout"] * t["cin"] * t["k"] * t["k"]
    return total


def num_params(scale: str | tuple, nc: int) -> int:
    """Parameter count of the un-fused model: conv weights + BatchNorm (gamma, beta) for Conv modules,
    weights + bias for the 6 plain Conv2d, + the DFL conv (16)."""
    n = 0
    for t in conv_table(scale, nc):
        n += t["cout"] * t["cin"] * t["k"] * t["k"]
        n += 2 * t["cout"] if t["act"] else t["cout"]
    return n + REG_MAX


# -------------------------------------------------------------------------------------------------
# synthetic weights (there are no trained weights: reference .MISSING_LARGE_BLOBS:6-7)
# -------------------------------------------------------------------------------------------------
_SILU_SECOND_MOMENT = 0.3557  # E[silu(z)^2], z ~ N(0,1)
_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def _stored_gains(scale, seed: int = 0) -> dict:
    """Per-conv scale factors measured once by tools/calibrate_synth_gains.py (63 floats per scale).  They belong to a weight DRAW:
    seed 0 uses synth_gain_<scale>.json, another seed its own synth_gain_<scale>_seed<k>.json where one was calibrated (scale s, seeds
    1-3) and the seed-0 table otherwise (with it a different draw may drift: seed 1 at scale s saturates every class score)."""
    if isinstance(scale, str):
        for name in ((f"synth_gain_{scale}_seed{seed}.json",) if seed else ()) + (f"synth_gain_{scale}.json",):
            path = os.path.join(_DATA_DIR, name)
            if os.path.exists(path):
                return json.load(open(path))
    return {}


def synthetic_weights(scale: str | tuple = "s", nc: int = 1, seed: int = 0, gains: dict | None = None) -> Dict[str, Tuple[np.ndarray, np.ndarray]]:
    """Seeded random detector weights, deterministic for (scale, nc, seed).

    Each conv is uniform(-b, b) with a variance-preserving bound for SiLU, times a stored per-conv
    gain that keeps pre-activations at unit scale through the whole net (a drifting net saturates
    every score and turns the arg-max parity test into a test of ties).  Detect biases follow
    ultralytics' `bias_init`: box branch 1.0, cls branch log(5 / nc / (640 / stride)^2), so scores
    sit sparsely around the conf = 0.1 threshold and both the "detection" and the "NaN row" outcomes
    occur."""
    rng = np.random.RandomState(seed)
    if gains is None:
        gains = _stored_gains(scale, seed)
    out = {}
    for t in conv_table(scale, nc):
        fan_in = t["cin"] * t["k"] * t["k"]
        nm = t["name"]
        if nm == "model.0":
            # inputs are pixels in [0,1] (mean ~0.75 for bright-field frames), not unit-variance
            std = 1.0 / math.sqrt(fan_in * 0.35)
        else:
            std = 1.0 / math.sqrt(fan_in * _SILU_SECOND_MOMENT)
        bound = math.sqrt(3.0) * std
        w = rng.uniform(-bound, bound, size=(t["cout"], t["k"], t["k"], t["cin"])).astype(np.float32)
        b = (0.1 * rng.standard_normal(t["cout"])).astype(np.float32)
        if nm.startswith("model.22.") and nm.endswith(".2"):
            lvl = int(nm.split(".")[3])
            if ".cv2." in nm:
                b = np.full(t["cout"], 1.0, dtype=np.float32) + (0.05 * rng.standard_normal(t["cout"])).astype(np.float32)
            else:
                b = np.full(t["cout"], math.log(5.0 / nc / (640.0 / STRIDES[lvl]) ** 2), dtype=np.float32)
        w *= np.float32(gains.get(nm, 1.0))
        out[nm] = (w, b)
    return out


MAGIC = b"WTKYOLO1"


def save_weights(path: str, weights: Dict[str, Tuple[np.ndarray, np.ndarray]], scale: str | tuple, nc: int):
    """Flat little-endian file: magic, nc, n_convs, then per conv: name[64], cout, cin, k, stride, act, W, b."""
    table = conv_table(scale, nc)
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<ii", nc, len(table)))
        for t in table:
            w, b = weights[t["name"]]
            f.write(t["name"].encode().ljust(64, b"\0"))
            f.write(struct.pack("<iiiii", t["cout"], t["cin"], t["k"], t["stride"], t["act"]))
            f.write(np.ascontiguousarray(w, dtype="<f4").tobytes())
            f.write(np.ascontiguousarray(b, dtype="<f4").tobytes())


def load_weights(path: str) -> tuple[Dict[str, Tuple[np.ndarray, np.ndarray]], int]:
    out = {}
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError(f"{path}: not a WTKYOLO1 weight file")
        nc, n = struct.unpack("<ii", f.read(8))
        for _ in range(n):
            name = f.read(64).rstrip(b"\0").decode()
            cout, cin, k, stride, act = struct.unpack("<iiiii", f.read(20))
            w = np.frombuffer(f.read(4 * cout * k * k * cin), dtype="<f4").reshape(cout, k, k, cin).copy()
            b = np.frombuffer(f.read(4 * cout), dtype="<f4").copy()
            out[name] = (w, b)
    return out, nc


def letterbox_shape(h: int, w: int, imgsz: int, stride: int = 32) -> tuple[int, int]:
    """Network input (H, W) ultralytics uses for a batch of same-shape frames with a .pt model
    (LetterBox auto=True: scale to fit imgsz, pad only up to the next stride multiple)."""
    r = min(imgsz / h, imgsz / w)
    new_w, new_h = int(round(w * r)), int(round(h * r))
    dw, dh = (imgsz - new_w) % stride, (imgsz - new_h) % stride
    return new_h + dh, new_w + dw
