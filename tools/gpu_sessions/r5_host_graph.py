"""Host entry point of a throughput-plan handle: eager launches (default) against the replayed capture (WTK_GRAPH_HOST=1), host-inclusive ms per call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wtracker_amd import hip, yolo_spec as ys, frames as fr

w = ys.synthetic_weights("s", 1, seed=0)
for dtype in ("f16x3", "fp16"):
    for B in (1, 4, 15):
        det = hip.HipYolo(w, (384, 384), 16, dtype=dtype, plan="throughput")
        f = fr.diverse_frames(16, 384, seed=1)[:B]
        for _ in range(5):
            det.predict_host(f, conf=0.1)
        t0 = time.perf_counter()
        n = 200
        for _ in range(n):
            det.predict_host(f, conf=0.1)
        dt = (time.perf_counter() - t0) / n
        print(f"WTK_GRAPH_HOST={os.environ.get('WTK_GRAPH_HOST', '0')} {dtype} B={B}: {dt * 1e3:.3f} ms per predict_host call", flush=True)
        det.close()
