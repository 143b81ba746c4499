import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from oracle import yolo_oracle as yo
from wtracker_amd import yolo_spec as ys, frames as fr
w = ys.synthetic_weights("s", 1, seed=0)
m = yo.YoloOracle(w, ys.model_dims(0.5, 0.33, 1024, 1))
f, _ = fr.synthetic_frames(16, 640, seed=1)
print("cores", os.cpu_count())
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    yo.predict(m, list(f[:2]), imgsz=640)
    t=time.perf_counter(); yo.predict(m, list(f), imgsz=640); dt=time.perf_counter()-t
    print(nt, "threads:", 16/dt, "frames/s", flush=True)
