"""Seeded synthetic bright-field microscopy frames (there are no real experiment images:
reference .MISSING_LARGE_BLOBS).  Spec from SURVEY.md §8d: background 200 + N(0,4) clipped, one
dark worm per frame (capsule body ~90 x 6 px with a ~14 x 14 px head, intensity 60), head moving as
a random walk with speed mean 0.54 px/frame, sigma 0.28, reflected at the borders.
"""
from __future__ import annotations

import numpy as np


def head_track(num_frames: int, size: int, seed: int) -> np.ndarray:
    """[N,3] (x, y, heading) of the head centre."""
    rng = np.random.default_rng(seed)
    pos = np.array([size * 0.5, size * 0.5]) + rng.uniform(-0.2, 0.2, 2) * size
    heading = rng.uniform(0, 2 * np.pi)
    out = np.empty((num_frames, 3))
    lo, hi = 30.0, size - 30.0
    for i in range(num_frames):
        speed = max(0.0, rng.normal(0.54, 0.28))
        heading += rng.normal(0.0, 0.12)
        pos = pos + speed * np.array([np.cos(heading), np.sin(heading)])
        for a in range(2):
            if pos[a] < lo:
                pos[a] = 2 * lo - pos[a]
                heading = np.pi - heading if a == 0 else -heading
            elif pos[a] > hi:
                pos[a] = 2 * hi - pos[a]
                heading = np.pi - heading if a == 0 else -heading
        out[i] = (pos[0], pos[1], heading)
    return out


def _draw_worm(img: np.ndarray, x: float, y: float, heading: float):
    """Dark capsule trailing behind the head + a rounder head blob (in place, uint8)."""
    H, W = img.shape
    body_len, body_r, head_r = 90.0, 3.0, 7.0
    tx, ty = x - body_len * np.cos(heading), y - body_len * np.sin(heading)
    x0, x1 = int(max(0, min(x, tx) - 10)), int(min(W, max(x, tx) + 11))
    y0, y1 = int(max(0, min(y, ty) - 10)), int(min(H, max(y, ty) + 11))
    if x1 <= x0 or y1 <= y0:
        return
    yy, xx = np.mgrid[y0:y1, x0:x1].astype(np.float64)
    dx, dy = tx - x, ty - y
    t = np.clip(((xx - x) * dx + (yy - y) * dy) / (dx * dx + dy * dy), 0.0, 1.0)
    d_body = np.hypot(xx - (x + t * dx), yy - (y + t * dy))
    d_head = np.hypot(xx - x, yy - y)
    mask = (d_body <= body_r) | (d_head <= head_r)
    sub = img[y0:y1, x0:x1]
    sub[mask] = 60


def synthetic_frames(num_frames: int, size: int = 640, seed: int = 0, colored: bool = False) -> tuple:
    """-> (frames uint8 [N,S,S] or [N,S,S,3] BGR, track [N,3])."""
    rng = np.random.default_rng(seed + 7919)
    track = head_track(num_frames, size, seed)
    frames = np.empty((num_frames, size, size), dtype=np.uint8)
    for i in range(num_frames):
        bg = np.clip(200.0 + rng.normal(0.0, 4.0, size=(size, size)), 0, 255)
        img = bg.astype(np.uint8)
        _draw_worm(img, *track[i])
        frames[i] = img
    if colored:
        frames = np.repeat(frames[..., None], 3, axis=3)
    return frames, track


def diverse_frames(num_frames: int, size: int = 640, seed: int = 1000, per_seed: int = 4) -> np.ndarray:
    """`num_frames` frames drawn from many tracks (`per_seed` consecutive frames per seed): consecutive frames of ONE
    synthetic track differ by half a pixel of worm motion only, which makes a poor sample for accuracy statistics."""
    out, n, s = [], 0, seed
    while n < num_frames:
        f, _ = synthetic_frames(min(per_seed, num_frames - n), size, seed=s)
        out.append(f)
        n += len(f)
        s += 1
    return np.concatenate(out)
