#!/usr/bin/env python3
"""bench.py — frames/s of the YOLOv8s + ResMLP sim-loop hot path on MI355X (BASELINE.json metric).

One "step" = one super-batch: every rank runs the detector (63 conv layers in ~50 launches: fused front,
fused C2f tail, implicit-GEMM and window-kernel convs with fused Detect tails, SPPF pool, head select) on its `--batch` synthetic 640x640 frames that are already resident in HBM, the [B,4]
track slices are all-gathered (N > 1 only), and the ResMLP movement vectors of the cycles that became
computable are produced (wtracker_amd/pipeline.py).  Prints ONE JSON line on rank 0.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_TFLOPS = {"fp16": 2500.0, "fp32": 157.3}  # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline(weights, dims, size: int, n_frames: int, batch: int, folded_path: str) -> dict:
    """The CPU restatement (oracle/, kind 'port') on a bounded sample of the same workload.  Thread count:
    the torch-CPU conv stack peaks near 32 threads on the GPU box's host (8: 14.8, 16: 18.3, 32: 19.2,
    64: 11.4, 128: 5.9 frames/s measured with tools/cpu_threads_probe.py), so min(cores, 32) is used."""
    from oracle import resmlp_oracle
    from oracle import yolo_oracle as yo
    from wtracker_amd import frames as fr

    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    model = yo.YoloOracle(weights, dims)
    frames, _ = fr.synthetic_frames(min(n_frames, 2 * batch), size, seed=1)
    st = resmlp_oracle.load_state(folded_path)
    yo.predict(model, list(frames[:2]), imgsz=size)  # warm-up
    t0 = time.perf_counter()
    done = 0
    while done < n_frames:
        b = min(batch, n_frames - done)
        off = done % max(len(frames) - b + 1, 1)
        yo.predict(model, list(frames[off : off + b]), imgsz=size)
        done += b
    # ResMLP over the sample's cycles (one sample per 9 frames)
    x = np.zeros((max(n_frames // 9, 1), 28), dtype=np.float32)
    resmlp_oracle.forward(st, x)
    dt = time.perf_counter() - t0
    return {"value": n_frames / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n_frames} synthetic {size}x{size} frames in batches of {batch}, torch-CPU fp32 restatement "
                      f"(oracle/yolo_oracle.py) + numpy ResMLP, {cores} threads, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU per step (BASELINE config 3: 64)")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "fp32"])
    ap.add_argument("--pool", type=int, default=128, help="distinct synthetic frames kept in HBM per rank")
    ap.add_argument("--cpu-frames", type=int, default=256, help="frames of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel-class HIP-event timing")
    ap.add_argument("--lanes", type=int, default=2, help="forward passes in flight per GPU (each lane = own workspace + HIP stream)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse the "
                    "multi-rank path on a one-GPU box, where every rank then shares cuda:0)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"WORLD_SIZE={world} != --gpus {args.gpus}")

    from wtracker_amd import _build, hip, resmlp
    from wtracker_amd import frames as fr
    from wtracker_amd import yolo_spec as ys
    from wtracker_amd.pipeline import TrackPipeline

    if world == 1 and _build.needs_build():
        _build.build(verbose=False)
    if args.backend != "nccl" and torch.cuda.device_count() <= local_rank:
        local_rank = 0  # rehearsal: ranks share the one visible GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    group = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        # only rank 0 (re)builds a stale library; nobody loads it before that is done
        if rank == 0 and _build.needs_build():
            _build.build(verbose=False)
        dist.barrier()
    if hip.device_count() <= local_rank:
        raise SystemExit("no HIP device visible: the product path has no CPU fallback")

    scale, nc = "s", 1
    weights = ys.synthetic_weights(scale, nc, seed=0)
    depth, width, maxch = ys.SCALES[scale]
    dets = [hip.HipYolo(weights, (args.size, args.size), args.batch, dtype=args.dtype, nc=nc, width=width, depth=depth,
                        max_channels=maxch, device=local_rank) for _ in range(args.lanes)]
    det = dets[0]
    golden = os.path.join(ROOT, "tests", "golden", "resmlp_100ms.npz")
    folded = resmlp.load_npz(golden)  # ResMLP(imaging-100ms_pred-40ms_moving-50ms): the reference's shipped weights
    mlp = hip.HipMLP(folded.layers, folded.n_blocks, folded.layers_per_block, device=local_rank)

    # synthetic frames, resident in HBM before the timed region; every rank draws its own seed
    pool = max(args.pool // args.batch, 1) * args.batch
    frames_np, _ = fr.synthetic_frames(pool, args.size, seed=1 + rank)
    frames = torch.from_numpy(frames_np).to(dev)
    n_pool_batches = pool // args.batch

    total_steps = args.warmup + args.steps
    total_frames = total_steps * args.batch * world
    # 60 fps, 100/40/50 ms timing (BASELINE config 3): imaging 6, pred 3, moving 3 frames
    pipe = TrackPipeline(dets, mlp, folded, args.batch, total_frames, imaging_frame_num=6, pred_frame_num=3, cycle_frame_num=9,
                         conf=0.1, rank=rank, world=world, group=group, device=dev)

    def run(s: int):
        b = s % n_pool_batches
        pipe.step(s, frames[b * args.batch : (b + 1) * args.batch])

    def fence():
        pipe.synchronize()
        torch.cuda.synchronize(dev)
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
            torch.cuda.synchronize(dev)

    for s in range(args.warmup):
        run(s)
    fence()
    t0 = time.perf_counter()
    for s in range(args.warmup, total_steps):
        run(s)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist

        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- per-kernel-class device time with HIP events on the launch stream (same workload, rank 0)
    roofline = None
    if not args.no_profile:
        fence()
        det.set_profiling(True)
        prof_steps = max(min(args.steps, 10), 1)
        for s in range(prof_steps):
            run((args.warmup + s) // args.lanes * args.lanes)  # lane 0 = the profiled handle
        fence()
        prof = det.get_profile()
        kprof = det.get_kernel_profile()
        det.set_profiling(False)
        peak = PEAK_TFLOPS[args.dtype]
        # HBM bytes per launch from the committed rocprofv3 PMC passes of this workload (FETCH_SIZE x2 + WRITE_SIZE,
        # tools/traffic_from_pmc.py); only valid for the configuration it was collected on
        tj = None
        tpath = os.path.join(ROOT, "profiles", "r01_conv_traffic.json")
        if os.path.exists(tpath) and args.size == 640 and args.dtype == "fp16":
            tj = json.load(open(tpath))

        uj = None
        upath = os.path.join(ROOT, "profiles", "r01_pmc_mfma_util.json")
        if os.path.exists(upath) and args.size == 640 and args.dtype == "fp16" and args.batch == 64:
            uj = json.load(open(upath))

        def kernel_line(name, k):
            avg_ms = k["total_ms"] / max(k["launches"], 1)
            flop_per_launch = k["flops"] / max(k["launches"], 1)
            ach = flop_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
            tr = None
            if tj and name in tj.get("per_kernel", {}):
                tr = tj["per_kernel"][name]["hbm_bytes_per_launch_avg"] * args.batch / tj["batch"]
            extra = {}
            if tr is not None and avg_ms > 0:  # HBM side of the same kernel: PMC bytes per launch / live launch duration, against ~8 TB/s
                extra["hbm_gbps"] = tr / (avg_ms * 1e-3) / 1e9
                extra["hbm_frac"] = extra["hbm_gbps"] / 8000.0
            if uj and name in uj["per_kernel"]:  # MFMA-busy share from the committed PMC pass (executed MFMAs, 2.4 GHz nominal)
                extra["mfma_util_pmc"] = uj["per_kernel"][name]["mfma_util"]
            return {"kernel": name, "achieved": ach, "frac": ach / peak, "traffic": tr, **extra, "launches_per_step": k["launches"] / prof_steps,
                    "avg_launch_ms": avg_ms, "flop_per_launch_avg": flop_per_launch, "ms_per_step": k["total_ms"] / prof_steps}

        conv_kernels = {n: k for n, k in kprof.items() if k["flops"] > 0 and k["launches"] > 0}
        lines = sorted((kernel_line(n, k) for n, k in conv_kernels.items()), key=lambda e: -e["ms_per_step"])
        dom = lines[0]  # the kernel the forward pass spends most of its device time in
        fam_ms = sum(k["total_ms"] for k in conv_kernels.values())
        fam_launches = sum(k["launches"] for k in conv_kernels.values())
        fam_flops = sum(k["flops"] for k in conv_kernels.values())
        fam_ach = fam_flops / (fam_ms * 1e-3) / 1e12
        roofline = {"kernel": dom["kernel"], "bound": "mfma", "achieved": dom["achieved"], "peak": peak, "unit": "TFLOP/s",
                    "frac": dom["frac"], "traffic": dom["traffic"], "hbm_gbps": dom.get("hbm_gbps"), "hbm_frac": dom.get("hbm_frac"),
                    "mfma_util_pmc": dom.get("mfma_util_pmc"), "launches_per_step": dom["launches_per_step"],
                    "avg_launch_ms": dom["avg_launch_ms"], "flop_per_launch_avg": dom["flop_per_launch_avg"],
                    "share_of_forward": dom["ms_per_step"] / sum(v["total_ms"] / prof_steps for v in prof.values()),
                    # every MFMA kernel of the forward pass together (what `value` is made of)
                    "conv_family": {"achieved": fam_ach, "frac": fam_ach / peak, "launches_per_step": fam_launches / prof_steps,
                                    "avg_launch_ms": fam_ms / fam_launches, "flop_per_launch_avg": fam_flops / fam_launches,
                                    "traffic": (tj["hbm_bytes_per_launch_avg"] * args.batch / tj["batch"]) if tj else None},
                    "kernels": lines,
                    "class_ms_per_step": {k: v["total_ms"] / prof_steps for k, v in prof.items()}}

    frames_done = args.steps * args.batch * world
    out = {
        "metric": f"frames/sec YOLOv8s+ResMLP sim loop @{args.size}x{args.size}",
        "value": frames_done / dt,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": ("BASELINE configs[2]" if args.size == 640 else f"BASELINE configs[4] per-GPU shape ({args.size}x{args.size})")
                               + ": full sim loop, YOLOv8s (nc=1, seeded synthetic weights) + ResMLP(imaging-100ms_pred-40ms_moving-50ms, reference weights)",
                   "frame": f"{args.size}x{args.size} uint8 gray, resident in HBM", "batch_per_gpu": args.batch, "lanes_per_gpu": args.lanes,
                   "global_batch": args.batch * world, "timing_ms": [100, 40, 50], "conf": 0.1,
                   "parallelism": f"frame-sharded x{world}, one RCCL all-gather of [B,4] tracks per step" if world > 1 else "single GPU"},
        "roofline": roofline,
    }
    if rank == 0 and world == 1 and args.cpu_frames > 0:  # the CPU baseline is reported at N=1 only
        out["cpu_baseline"] = cpu_baseline(weights, ys.model_dims(width, depth, maxch, nc), args.size, args.cpu_frames, args.batch, golden)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
