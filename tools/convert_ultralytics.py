#!/usr/bin/env python3
"""Offline converter: an ultralytics YOLOv8 detection checkpoint's state dict -> WTKYOLO1 weight file
(SURVEY.md §8 f3).  Run where `ultralytics` is installed (to unpickle a real `yolov8s_trained.pt`), or on a plain
state-dict file (`torch.save(model.state_dict(), path)`), which needs torch only.  No real checkpoint exists in this build
pipeline (no ultralytics, no trained weights): it is covered by the synthetic round trips
tests/test_oracle_yolo.py::test_converter_folding_round_trip (fold arithmetic, CPU) and
tests/test_gpu_yolo.py::test_converted_unfused_checkpoint_runs_on_device (un-fused state dict -> this tool -> .wtk ->
HipYoloController against an oracle that applies conv -> BatchNorm(eps 1e-3) -> SiLU explicitly).

    python tools/convert_ultralytics.py yolov8s_trained.pt yolov8s_worm.wtk --scale s

Mapping (ultralytics module paths): `model.{i}.conv.weight` + `model.{i}.bn.{weight,bias,running_mean,
running_var}` (eps 1e-3) for every `Conv`; nested `cv1/cv2/m.{j}.cv1/...`; Detect towers
`model.22.cv2.{l}.{0,1}.conv/.bn`, plain `model.22.cv2.{l}.2.{weight,bias}`; same for `cv3`.
The DFL conv (`model.22.dfl.conv.weight` = arange(16)) is fixed and not stored.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wtracker_amd import yolo_spec as ys  # noqa: E402

BN_EPS = 1e-3  # ultralytics Conv: nn.BatchNorm2d(c2, eps=0.001, momentum=0.03)


def fold_state_dict(sd: dict, scale: str, nc: int) -> dict:
    """sd: name -> numpy array (OIHW conv weights).  Returns name -> (W O-H-W-I fp32, b fp32)."""
    out = {}
    for t in ys.conv_table(scale, nc):
        nm = t["name"]
        if t["act"]:
            w = np.asarray(sd[nm + ".conv.weight"], dtype=np.float64)
            g = np.asarray(sd[nm + ".bn.weight"], dtype=np.float64)
            beta = np.asarray(sd[nm + ".bn.bias"], dtype=np.float64)
            mu = np.asarray(sd[nm + ".bn.running_mean"], dtype=np.float64)
            var = np.asarray(sd[nm + ".bn.running_var"], dtype=np.float64)
            s = g / np.sqrt(var + BN_EPS)
            w = w * s[:, None, None, None]
            b = beta - mu * s
        else:
            w = np.asarray(sd[nm + ".weight"], dtype=np.float64)
            b = np.asarray(sd[nm + ".bias"], dtype=np.float64)
        assert w.shape == (t["cout"], t["cin"], t["k"], t["k"]), (nm, w.shape)
        out[nm] = (np.ascontiguousarray(w.transpose(0, 2, 3, 1)).astype(np.float32), b.astype(np.float32))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("checkpoint")
    ap.add_argument("output")
    ap.add_argument("--scale", default="s", choices=list(ys.SCALES))
    args = ap.parse_args(argv)
    import torch

    ck = torch.load(args.checkpoint, map_location="cpu", weights_only=False)
    if isinstance(ck, dict) and ck and all(torch.is_tensor(v) for v in ck.values()):
        sd = {k: v.detach().float().cpu().numpy() for k, v in ck.items()}  # a plain state dict
    else:
        model = ck["model"] if isinstance(ck, dict) and "model" in ck else ck
        model = model.float()
        sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    nc = int(sd["model.22.cv3.0.2.weight"].shape[0])
    ys.save_weights(args.output, fold_state_dict(sd, args.scale, nc), args.scale, nc)
    print(f"wrote {args.output}: scale {args.scale}, nc {nc}")


if __name__ == "__main__":
    main()
