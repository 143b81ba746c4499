#!/usr/bin/env python3
"""Device time (HIP events on the call's stream, 40 back-to-back calls) and host wall time per call of small-batch detector calls:
  python tools/gpu_sessions/r5_lat_time.py [dtype=f16x3] [plans=latency,throughput]      (WTK_GRAPH_MAX_BATCH=0: eager launches)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from wtracker_amd import hip, yolo_spec as ys, frames as fr

dtypes = (sys.argv[1] if len(sys.argv) > 1 else "f16x3").split(",")
plans = (sys.argv[2] if len(sys.argv) > 2 else "latency,throughput").split(",")
w = ys.synthetic_weights("s", 1, seed=0)
depth, width, maxch = ys.SCALES["s"]
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev)
for dtype in dtypes:
    for plan in plans:
        for size, B in ((384, 1), (384, 15), (640, 1)):
            det = hip.HipYolo(w, (size, size), 16, dtype=dtype, width=width, depth=depth, max_channels=maxch, plan=plan)
            f = torch.from_numpy(fr.diverse_frames(16, size, seed=1)[:B]).to(dev)
            x = torch.empty((B, 4), dtype=torch.float32, device=dev)
            call = lambda: det.predict(f, B, size, size, 1, x, conf=0.1, stream=st.cuda_stream)
            for _ in range(5):
                call()
            st.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(40):
                call()
            e1.record(st)
            st.synchronize()
            host = []
            for _ in range(30):
                t0 = time.perf_counter(); call(); st.synchronize(); host.append(time.perf_counter() - t0)
            print(f"{dtype} {det.plan:10s} {size} B={B:2d}: device {e0.elapsed_time(e1) / 40 * 1e3:8.1f} us/call   host {np.median(host) * 1e6:8.1f} us/call   graph_max={os.environ.get('WTK_GRAPH_MAX_BATCH', '16')}", flush=True)
            det.close()
