"""Child process of tests/test_a_gpu_two_ranks.py::test_rccl_collectives_in_a_one_rank_world: the torch.distributed calls of the N > 1 path — eager
`device_id=` initialisation of the RCCL communicator, barrier, all_reduce(MAX / MIN / SUM) of CUDA tensors (bench.py's window maximum and checksums),
`all_gather_into_tensor` into a view of the device-resident track issued on a lane stream (wtracker_amd.pipeline.exchange_tracks) — in a world of ONE rank
on the one visible GPU.  RCCL itself runs (communicator, kernels, streams); only the peer is missing."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from wtracker_amd.pipeline import ShardPlan, exchange_tracks  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    dist.barrier()
    B, steps = 32, 5
    plan = ShardPlan(B, 1, B * steps, 6, 3, 9)
    track = torch.full((B * steps, 4), float("nan"), dtype=torch.float32, device=dev)
    lanes = [torch.cuda.Stream(device=dev) for _ in range(2)]
    rows = [torch.randn((B, 4), dtype=torch.float32, device=dev) for _ in range(steps)]
    torch.cuda.synchronize(dev)
    for s in range(steps):  # as TrackPipeline._finalize_lane: on the lane's stream, its stream handle handed down
        with torch.cuda.stream(lanes[s % 2]):
            exchange_tracks(track, rows[s], plan, s, None, None, lanes[s % 2].cuda_stream)
    for st in lanes:
        st.synchronize()
    torch.cuda.synchronize(dev)
    assert torch.equal(track, torch.cat(rows)), "all_gather_into_tensor did not land the rows in frame order"
    t = torch.tensor([1.5, 0.25, 3.0], dtype=torch.float64, device=dev)
    for op in (dist.ReduceOp.MAX, dist.ReduceOp.MIN, dist.ReduceOp.SUM):
        u = t.clone()
        dist.all_reduce(u, op=op)
        assert torch.equal(u, t)
    dist.barrier()
    dist.destroy_process_group()
    print("rccl one-rank world ok")


if __name__ == "__main__":
    main()
