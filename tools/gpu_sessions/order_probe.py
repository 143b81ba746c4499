#!/usr/bin/env python3
"""Does a mode's rate depend on what ran before it in the same process?  Builds the bench's Workload for a sequence of modes and
times each (three windows of 20 steps, two lanes).   python tools/gpu_sessions/order_probe.py hybrid fp16 hybrid f16x3 hybrid"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from wtracker_amd import frames as fr  # noqa: E402
from wtracker_amd import resmlp  # noqa: E402
from wtracker_amd import yolo_spec as ys  # noqa: E402


def main():
    modes = sys.argv[1:] or ["hybrid", "fp16", "hybrid"]
    args = argparse.Namespace(size=640, batch=64, conf=0.1)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    weights = ys.synthetic_weights("s", 1, seed=0)
    depth, width, maxch = ys.SCALES["s"]
    folded = resmlp.load_npz(os.path.join(ROOT, "tests", "golden", "resmlp_100ms.npz"))
    frames = torch.from_numpy(fr.diverse_frames(128, 640, seed=3000)).to(dev)
    pre = int(os.environ.get("PREALLOC_GB", "0"))
    hold = [torch.empty(1 << 30, dtype=torch.uint8, device=dev).fill_(1) for _ in range(pre)]  # occupy the first GBs of the device heap
    print(f"preallocated {len(hold)} GiB", flush=True)
    lane_streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    for mode in modes:
        steps, reps = 20, 3
        n_steps = 5 + reps * steps + 2
        wl = bench.Workload(args, mode, 2, weights, (width, depth, maxch), folded, 0, 0, 1, None, dev, n_steps, streams=lane_streams)
        s = 0
        for _ in range(5):
            wl.pipe.step(s, frames[(s % 2) * 64 : (s % 2 + 1) * 64]); s += 1
        wl.pipe.synchronize(); torch.cuda.synchronize(dev)
        win = []
        for _ in range(reps):
            t0 = time.perf_counter()
            for _ in range(steps):
                wl.pipe.step(s, frames[(s % 2) * 64 : (s % 2 + 1) * 64]); s += 1
            wl.pipe.synchronize(); torch.cuda.synchronize(dev)
            win.append(time.perf_counter() - t0)
        print(f"{mode:8s} {steps * 64 / np.median(win):8.0f} frames/s  windows {[round(w * 1e3, 1) for w in win]}", flush=True)
        wl.close()
        del wl


if __name__ == "__main__":
    main()
