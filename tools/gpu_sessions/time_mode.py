#!/usr/bin/env python3
"""Time one precision mode of the detector on device-resident frames (single stream and, with --lanes 2, two forward passes in flight).

  python tools/gpu_sessions/time_mode.py --dtype f16x3 --batch 64 --size 640
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wtracker_amd import frames as fr  # noqa: E402
from wtracker_amd import hip  # noqa: E402
from wtracker_amd import yolo_spec as ys  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f16x3")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--lanes", type=int, default=1)
    ap.add_argument("--kernels", action="store_true", help="print the per-kernel-class profile of one forward")
    ap.add_argument("--plan", default="auto", choices=["auto", "throughput", "latency"])
    ap.add_argument("--dyn", type=int, default=-1, help="device-side dynamic batch count (wtk_yolo_set_dynamic_batch); -1: static")
    args = ap.parse_args()
    B, S = args.batch, args.size
    w = ys.synthetic_weights("s", 1, seed=0)
    depth, width, maxch = ys.SCALES["s"]
    dets = [hip.HipYolo(w, (S, S), B, dtype=args.dtype, nc=1, width=width, depth=depth, max_channels=maxch, plan=args.plan) for _ in range(args.lanes)]
    frames, _ = fr.synthetic_frames(B, S, seed=0)
    dev = torch.from_numpy(np.ascontiguousarray(frames)).cuda()
    outs = [torch.empty((B, 4), dtype=torch.float32, device="cuda") for _ in dets]
    streams = [torch.cuda.Stream() for _ in dets]
    if args.dyn >= 0:
        n_dev = torch.tensor([args.dyn], dtype=torch.int32, device="cuda")
        for d in dets:
            d.set_dynamic_batch(n_dev)

    def step():
        for d, o, s in zip(dets, outs, streams):
            d.predict(dev.data_ptr(), B, S, S, 1, o.data_ptr(), conf=0.1, stream=s.cuda_stream)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(f"{args.dtype} plan={dets[0].plan} B={B} dyn={args.dyn} {S}x{S} lanes={args.lanes}: {dt * 1e3:.3f} ms per step, {B * args.lanes / dt:,.0f} frames/s", flush=True)
    if args.kernels:
        dets[0].set_profiling(True)
        dets[0].predict(dev.data_ptr(), B, S, S, 1, outs[0].data_ptr(), conf=0.1, stream=0)
        torch.cuda.synchronize()
        print(dets[0].get_kernel_profile())


if __name__ == "__main__":
    main()
