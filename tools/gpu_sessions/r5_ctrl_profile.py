#!/usr/bin/env python3
"""Where a controller call's host time goes: cProfile over the closed loop (plan auto, f16x3), top functions by own time."""
import cProfile, os, pstats, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from harness.sim_harness import ArrayReader, Simulator
from wtracker_amd import frames as fr, yolo_spec as ys
from wtracker_amd.controllers import HipYoloController, YoloConfig
from wtracker_amd.sim import ExperimentConfig, TimingConfig, TrackLogger

size, cycles = 1024, 40
ec = ExperimentConfig("closed_loop", cycles * 15 + 1, 60, (size, size), 90, (size // 2, size // 2))
frames_np, _ = fr.synthetic_frames(ec.num_frames, size, seed=77)
dev_frames = torch.from_numpy(frames_np).cuda()
tmp = tempfile.NamedTemporaryFile(suffix=".wtk", delete=False); tmp.close()
ys.save_weights(tmp.name, ys.synthetic_weights("s", 1, seed=0), "s", 1)
cfg = YoloConfig(model_path=tmp.name, device="cuda:0", pred_kwargs={"imgsz": 384, "conf": 0.1}, dtype="f16x3", scale="s", max_batch=16)
tc = TimingConfig(ec, 200, 40, 50, (4, 4), (0.32, 0.32))
ctrl = HipYoloController(tc, cfg, device_frames=dev_frames)
Simulator(tc, ec, TrackLogger(ctrl), reader=ArrayReader(frames_np)).run()  # warm-up run: handles, buffers, captures
ctrl2 = HipYoloController(tc, cfg, device_frames=dev_frames)
pr = cProfile.Profile()
pr.enable()
Simulator(tc, ec, TrackLogger(ctrl2), reader=ArrayReader(frames_np)).run()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
os.unlink(tmp.name)
