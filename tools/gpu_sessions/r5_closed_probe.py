#!/usr/bin/env python3
"""closed-loop call times, every call listed: python tools/gpu_sessions/r5_closed_probe.py [plan=auto] [deferred]"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from harness.sim_harness import ArrayReader, Simulator
from wtracker_amd import frames as fr, yolo_spec as ys
from wtracker_amd.controllers import HipYoloController, YoloConfig
from wtracker_amd.sim import ExperimentConfig, TimingConfig, TrackLogger

plan = sys.argv[1] if len(sys.argv) > 1 else "auto"
deferred = len(sys.argv) > 2 and sys.argv[2] == "deferred"
size, cycles = 1024, 10
ec = ExperimentConfig("closed_loop", cycles * 15 + 1, 60, (size, size), 90, (size // 2, size // 2))
frames_np, _ = fr.synthetic_frames(ec.num_frames, size, seed=77)
dev_frames = torch.from_numpy(frames_np).cuda()
tmp = tempfile.NamedTemporaryFile(suffix=".wtk", delete=False); tmp.close()
ys.save_weights(tmp.name, ys.synthetic_weights("s", 1, seed=0), "s", 1)
cfg = YoloConfig(model_path=tmp.name, device="cuda:0", pred_kwargs={"imgsz": 384, "conf": 0.1}, dtype="f16x3", scale="s", max_batch=16, plan=plan)
for rep in range(3):
    tc = TimingConfig(ec, 200, 40, 50, (4, 4), (0.32, 0.32))
    ctrl = HipYoloController(tc, cfg, device_frames=dev_frames)
    calls = []
    inner = ctrl.predict_views
    def timed(e, _i=inner):
        t0 = time.perf_counter(); r = _i(e); calls.append((len(e), (time.perf_counter() - t0) * 1e3)); return r
    ctrl.predict_views = timed
    t0 = time.perf_counter()
    Simulator(tc, ec, TrackLogger(ctrl, deferred=deferred), reader=ArrayReader(frames_np)).run()
    dt = time.perf_counter() - t0
    print(f"plan {plan} rep {rep}: {ec.num_frames / dt:.0f} frames/s, {dt / cycles * 1e3:.3f} ms per cycle; calls (B: ms): " + " ".join(f"{b}:{t:.2f}" for b, t in calls), flush=True)
os.unlink(tmp.name)
