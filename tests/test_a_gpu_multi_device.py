"""GPU, self-arming: the multi-rank path on REAL devices (the batch map that is sharded is
yolo_controller.py:108-109).  Skipped on a one-GPU box; on a node with >= 2 visible MI355X `pytest -m gpu` runs

  (i)  tests/dist_worker.py --backend nccl, one GPU per rank: torch.distributed's RCCL communicator, `device_id=` eager init,
       `all_gather_into_tensor` into a view of the device-resident track from two lane streams per rank;
  (ii) the same pipeline with `TrackPipeline(comm=hip.WtkComm(...))`: the C ABI's own RCCL communicator
       (wtk_comm_unique_id / wtk_comm_create / wtk_allgather_tracks), rendezvous token handed over through a file;

and every rank must end with the track, validity flags and ResMLP moves of a single-rank run over the same frames, bit for bit
(what the gloo rehearsal on one device, test_a_gpu_two_ranks.py, checks without RCCL).

The rank processes are fresh children started BEFORE this process touches the GPU (conftest.py keeps test_a_gpu_* first;
`torch.cuda.device_count()` does not initialise HIP on this image); nothing is re-exec'ed."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_devices() -> int:
    import torch

    return torch.cuda.device_count()


def _spawn(world, tmp_path, tag, extra, batch=32, backend="nccl"):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), "--out", str(tmp_path / f"{tag}{r}.npz"),
               "--batch", str(batch), "--steps", "4", "--lanes", "2" if world > 1 else "1", "--backend", backend] + extra
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"{tag} rank {r} failed:\n{outs[r][-3000:]}"


@pytest.mark.skipif(_n_devices() < 2, reason="needs >= 2 visible GPUs (RCCL refuses two ranks on one device)")
def test_rccl_ranks_on_their_own_devices_equal_single_rank(hip_lib, tmp_path):
    import torch

    assert not torch.cuda.is_initialized(), "rank processes must be started before this process touches the GPU"
    world = min(_n_devices(), 4)
    _spawn(world, tmp_path, "torch_nccl", [])
    _spawn(world, tmp_path, "wtk_comm", ["--wtkcomm", str(tmp_path / "rccl_token.bin")])
    # the single-rank comparison run is a child process as well (one rank, the whole super-batch per step): this process never
    # initialises the GPU here, so the one-device rehearsal that follows (test_a_gpu_two_ranks.py) can still start its children
    _spawn(1, tmp_path, "single", [], batch=32 * world, backend="gloo")
    assert not torch.cuda.is_initialized()
    one = np.load(tmp_path / "single0.npz")
    assert np.isfinite(one["track"]).any() and one["valid"].sum() >= 10
    for transport in ("torch_nccl", "wtk_comm"):
        for r in range(world):
            z = np.load(tmp_path / f"{transport}{r}.npz")
            np.testing.assert_array_equal(z["track"], one["track"], err_msg=f"{transport} rank {r}")
            np.testing.assert_array_equal(z["valid"], one["valid"])
            np.testing.assert_array_equal(z["moves"], one["moves"])


def test_multi_device_tests_are_armed_or_skipped_for_the_stated_reason():
    """Always runs: records on a one-GPU box that the test above exists and why it did not run."""
    n = _n_devices()
    assert n >= 1
    print(f"\nvisible GPUs: {n} -> RCCL multi-device test {'ARMED' if n >= 2 else 'skipped (one device)'}")
