import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from wtracker_amd import hip, resmlp
for tag in ("100", "200"):
    m = resmlp.load_npz(f"tests/golden/resmlp_{tag}ms.npz")
    h = hip.HipMLP(m.layers, m.n_blocks, m.layers_per_block)
    for B in (1, 7, 256):
        x = torch.randn(B, m.in_dim, device="cuda")
        y = torch.empty(B, m.out_dim, device="cuda")
        for _ in range(10): h.forward(x, y, B)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): h.forward(x, y, B)
        e1.record(); torch.cuda.synchronize()
        print(tag, "B", B, round(e0.elapsed_time(e1) / 200 * 1e3, 2), "us per forward")
