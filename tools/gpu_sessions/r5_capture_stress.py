"""Host-side stress of the first-call capture (create handle -> one predict_host that captures and replays a hipGraph -> destroy), the call an intermittent
host SIGSEGV was seen in twice.  WTK_SEGV_BACKTRACE=1 prints the native frames if it happens again.  Usage: r5_capture_stress.py [iterations] [dtype]"""
import faulthandler
import os
import sys
import time

faulthandler.enable()
os.environ.setdefault("WTK_SEGV_BACKTRACE", "1")
os.environ.setdefault("WTK_GRAPH_HOST", "1")
os.environ.setdefault("WTK_LATENCY_PLAN", "0")
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wtracker_amd import hip, yolo_spec as ys

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dtypes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["fp32", "fp16"]
H, W = 96, 160
w = ys.synthetic_weights("n", 1, seed=0)
rng = np.random.default_rng(3)
frames = rng.integers(0, 256, size=(3, H, W), dtype=np.uint8)
t0 = time.time()
tt = [torch.zeros(1 << 20, device="cuda")]  # torch's context and allocator live in the process, as in a test run
keep = []
for i in range(n):
    dt = dtypes[i % len(dtypes)]
    det = hip.HipYolo(w, (H, W), 3, dtype=dt, nc=1, width=0.25, depth=0.33, max_channels=1024)
    det.predict_host(frames, conf=0.05)
    if i % 3 == 0:
        det.predict_host(frames[:1], conf=0.05)
    keep.append(det)  # destruction out of phase with creation, as the garbage collector does it in a test run
    if len(keep) > (i % 4):
        keep.pop(0).close()
    if i % 7 == 0:
        tt.append(torch.randn(1 << (10 + i % 12), device="cuda"))
        if len(tt) > 5:
            tt.pop(0)
    if i % 100 == 99:
        print(f"{i + 1} handles, {time.time() - t0:.0f} s", flush=True)
print("no crash")
