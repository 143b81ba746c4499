"""The multi-GPU path's only collective (all-gather of [B,4] track slices) and the interleaved
super-batch sharding, exercised with 2 CPU ranks over gloo: every rank must end with the same track
as a single-process run, and the per-step cycle sets must partition all cycles.  CPU only."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from wtracker_amd.pipeline import ShardPlan, exchange_tracks


def fake_detect(frame_idx: np.ndarray) -> np.ndarray:
    """Deterministic stand-in for the detector: a box that depends only on the global frame index."""
    f = frame_idx.astype(np.float32)
    out = np.stack([100 + 0.5 * f, 200 - 0.25 * f, 14 + (f % 3), 15 + (f % 5)], axis=1).astype(np.float32)
    out[frame_idx % 37 == 5] = np.nan  # missed detections travel as NaN rows
    return out


def _worker(rank, world, port, B, steps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan = ShardPlan(B, world, steps * B * world, imaging_frame_num=6, pred_frame_num=3, cycle_frame_num=9)
    track = torch.full((plan.total_frames, 4), float("nan"))
    cycles = []
    for s in range(plan.steps):
        f0, f1 = plan.local_range(s, rank)
        local = torch.from_numpy(fake_detect(np.arange(f0, f1)))
        exchange_tracks(track, local, plan, s)
        lo, hi = plan.cycles(s)
        # after step s every frame the cycles look back at (<= 27 frames) is present
        for a in plan.anchors[lo:hi]:
            need = a + np.array([0, -2, -9, -11, -18, -20, -27])
            assert need.max() < plan.super_range(s)[1]
        cycles.append((lo, hi))
    q.put((rank, track.numpy(), cycles))
    dist.barrier()
    dist.destroy_process_group()


def _run_world(world, B, steps):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = steps * B * world
    expect = fake_detect(np.arange(total))
    for rank, track, cycles in res:
        np.testing.assert_array_equal(track, expect)  # NaN rows compare equal position-wise
        assert cycles[0][0] == 0 and all(cycles[i][1] == cycles[i + 1][0] for i in range(len(cycles) - 1))
    plan = ShardPlan(B, world, total, 6, 3, 9)
    assert res[0][2][-1][1] == len(plan.anchors)


def test_two_rank_gloo_allgather_matches_single_process():
    _run_world(2, 8, 5)


@pytest.mark.parametrize("world,B,steps", [(3, 5, 4), (4, 8, 3)])
def test_three_and_four_rank_gloo_allgather_matches_single_process(world, B, steps):
    """The rank count of the driver's scaling runs is 1, 2, 4, 8: the interleaved super-batches and the gather must not assume two ranks
    (odd world size and a batch that is no power of two included)."""
    _run_world(world, B, steps)


def test_shard_plan_single_rank_degenerate_case():
    plan = ShardPlan(64, 1, 640, 6, 3, 9)
    assert plan.steps == 10 and plan.local_range(3, 0) == plan.super_range(3) == (192, 256)
    assert plan.anchors[0] == 3 and (np.diff(plan.anchors) == 9).all()
    seen = []
    for s in range(plan.steps):
        lo, hi = plan.cycles(s)
        seen += list(range(lo, hi))
    assert seen == list(range(len(plan.anchors)))


def test_shard_plan_50k_frames_over_8_ranks():
    """BASELINE config 4 geometry (index arithmetic only): 50 000 frames, 8 ranks x 64 frames per step."""
    plan = ShardPlan(64, 8, 50_000 // 512 * 512, 6, 3, 9)
    assert plan.super_batch == 512 and plan.steps == 97
    covered = np.zeros(plan.total_frames, dtype=np.int32)
    for s in range(plan.steps):
        for r in range(8):
            f0, f1 = plan.local_range(s, r)
            covered[f0:f1] += 1
        lo, hi = plan.cycles(s)
        if hi > lo:  # every look-back row of these cycles is inside the gathered prefix
            assert plan.anchors[hi - 1] + plan.pred < plan.super_range(s)[1]
    assert (covered == 1).all()
    assert plan.cycles(plan.steps - 1)[1] == len(plan.anchors) == (plan.total_frames - 6) // 9 + 1
