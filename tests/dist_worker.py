"""Rank process of tests/test_a_gpu_two_ranks.py (not a test module): runs wtracker_amd.pipeline.TrackPipeline as one of
WORLD_SIZE ranks over `--backend` (gloo: every rank shares cuda:0 on a one-GPU box; nccl: one GPU per rank) and writes its
view of the run (track, moves, valid) to --out.  Launched as a fresh child before anything in the parent touches the GPU."""
from __future__ import annotations

import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


import numpy as np  # noqa: E402


def run_pipeline(frames_np, batch, steps, rank, world, group, device, lanes, scale="n", size=128, dtype="fp16", hybrid=False, comm=None, defer=1):
    """Shared by the rank processes and by the single-rank comparison run in the test.  `hybrid`: scale-s HybridDetector lanes (fp16 +
    an f16x3 second look at every row below the margin: ceiling = batch, so the replaced rows do not depend on how frames are batched)."""
    import torch

    from wtracker_amd import hip, resmlp
    from wtracker_amd import yolo_spec as ys
    from wtracker_amd.pipeline import TrackPipeline

    folded = resmlp.load_npz(os.path.join(ROOT, "tests", "golden", "resmlp_100ms.npz"))
    if hybrid:
        scale = "s"
    w = ys.synthetic_weights(scale, 1, seed=0)
    depth, width, maxch = ys.SCALES[scale]

    def handle(dt):
        return hip.HipYolo(w, (size, size), batch, dtype=dt, nc=1, width=width, depth=depth, max_channels=maxch, device=device.index or 0)

    if hybrid:
        from wtracker_amd.hybrid import HybridDetector

        # wide margin: plenty of rows take the second look; defer > 1: the weak rows of `defer` steps of a lane share one full-precision pass, the
        # lane's all-gathers and the ResMLP of those steps follow the flush (wtracker_amd/pipeline.py)
        big = lambda: hip.HipYolo(w, (size, size), batch * defer, dtype="f16x3", nc=1, width=width, depth=depth, max_channels=maxch, device=device.index or 0)
        dets = [HybridDetector(handle("fp16"), big(), margin=0.5, k=batch * defer, defer=defer) for _ in range(lanes)]
    else:
        dets = [handle(dtype) for _ in range(lanes)]
    mlp = hip.HipMLP(folded.layers, folded.n_blocks, folded.layers_per_block, device=device.index or 0)
    total = steps * batch * world
    pipe = TrackPipeline(dets, mlp, folded, batch, total, imaging_frame_num=6, pred_frame_num=3, cycle_frame_num=9, conf=0.1,
                         rank=rank, world=world, group=group, device=device, comm=comm)
    frames = torch.from_numpy(frames_np).to(device)
    for s in range(steps):
        f0, f1 = pipe.plan.local_range(s, rank)
        pipe.step(s, frames[f0:f1])
    pipe.synchronize()
    torch.cuda.synchronize(device)
    out = dict(track=pipe.track.cpu().numpy(), moves=pipe.moves.cpu().numpy(), valid=pipe.valid.cpu().numpy())
    if hybrid:
        out["replaced"] = np.array([sum(int(d.replaced.item()) for d in dets)])
    for d in dets:
        d.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--lanes", type=int, default=2)
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--hybrid", action="store_true")
    ap.add_argument("--defer", type=int, default=1)
    ap.add_argument("--wtkcomm", default="", help="rendezvous file: exchange tracks through the C ABI's own RCCL communicator (hip.WtkComm) instead of torch.distributed")
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])

    import numpy as np
    import torch
    import torch.distributed as dist

    from wtracker_amd import frames as fr

    local = int(os.environ.get("LOCAL_RANK", rank))
    dev = torch.device("cuda", local if args.backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if args.wtkcomm:
        # no torch.distributed at all: rank 0 writes the 128-byte RCCL token to a file (the out-of-band hand-over wtk_hip.h describes), the others pick it up
        import time

        from wtracker_amd import hip

        if rank == 0:
            with open(args.wtkcomm + ".tmp", "wb") as f:
                f.write(hip.comm_unique_id())
            os.replace(args.wtkcomm + ".tmp", args.wtkcomm)
        t0 = time.time()
        while not os.path.exists(args.wtkcomm):
            if time.time() - t0 > 120:
                raise SystemExit("rendezvous token never appeared")
            time.sleep(0.05)
        comm = hip.WtkComm(dev.index, rank, world, open(args.wtkcomm, "rb").read())
        frames_np, _ = fr.synthetic_frames(args.steps * args.batch * world, 128, seed=4)
        out = run_pipeline(frames_np, args.batch, args.steps, rank, world, None, dev, args.lanes, hybrid=args.hybrid, comm=comm, defer=args.defer)
        np.savez(args.out, **out)
        comm.close()
        return
    if args.backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    frames_np, _ = fr.synthetic_frames(args.steps * args.batch * world, 128, seed=4)
    out = run_pipeline(frames_np, args.batch, args.steps, rank, world, None, dev, args.lanes, hybrid=args.hybrid, defer=args.defer)
    np.savez(args.out, **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
