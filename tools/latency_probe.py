#!/usr/bin/env python3
"""Latency of the controller-style host entry point (wtk_yolo_predict_host) for small batches, with and
without hipGraph replay (WTK_GRAPH_MAX_BATCH=0 disables)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wtracker_amd import hip, yolo_spec as ys, frames as fr

size = int(sys.argv[1]) if len(sys.argv) > 1 else 384
dtypes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["fp16"]
w = ys.synthetic_weights("s", 1, seed=0)
f, _ = fr.synthetic_frames(16, size, seed=0)
for dtype in dtypes:
    det = hip.HipYolo(w, (size, size), 16, dtype=dtype)
    for B in (1, 9, 15):
        for _ in range(5):
            det.predict_host(f[:B])
        t = time.perf_counter()
        n = 50
        for _ in range(n):
            det.predict_host(f[:B])
        dt = (time.perf_counter() - t) / n
        print(f"{dtype} size {size} B={B}: {dt*1e3:.3f} ms/call  ({B/dt:.0f} frames/s)  graph_max={os.environ.get('WTK_GRAPH_MAX_BATCH','16')}", flush=True)
    det.close()
