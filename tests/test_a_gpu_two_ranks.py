"""GPU: the multi-rank branch of wtracker_amd.pipeline.TrackPipeline as a driver-run test.

Two rank processes (gloo rendezvous on 127.0.0.1, both on cuda:0 — a one-GPU box cannot run RCCL between two devices) run
the REAL pipeline: detector on each rank's interleaved share, `all_gather_into_tensor` of the [B,4] slices into the
device-resident track from two lane streams, ResMLP over the cycles that became computable.  Every rank must end with the
track, validity flags and moves of a single-rank run over the same frames.

The children are started as fresh processes BEFORE this process has touched the GPU (conftest.py moves this module to
the front of the collection, and the test checks that nothing initialised HIP yet); the comparison run follows."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_n_gt_1_branch_with_two_gloo_ranks(hip_lib):
    """bench.py's N > 1 branch as a driver-run test (until now it had only been run by hand): `bench.py --gpus 2 --backend gloo`
    as two rank processes sharing cuda:0 (started before this process touches the GPU).  The printed line must say world_size 2, carry a track
    checksum that is equal on both ranks, headline the reference-precision mode with its device-side check green, and its `value` must be the
    frames BOTH ranks processed in a window over that window's time."""
    import json

    import torch

    assert not torch.cuda.is_initialized(), "this test must run before anything touches the GPU in this process"
    world, steps = 2, 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("GPU_MAX_HW_QUEUES", None)  # the ranks share one device: the runtime default
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--backend", "gloo", "--steps", str(steps), "--warmup", "1", "--repeats", "1",
               "--no-fp32", "--cpu-frames", "0", "--pool", "64"]
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][1][-3000:]}"
    assert not torch.cuda.is_initialized()
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")], "only rank 0 prints the line"
    raw = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(raw) == 1 and outs[0][0].rstrip().endswith(raw[0]), "ONE json line, and it is the last thing on stdout"
    assert len(raw[0]) < 4096, len(raw[0])  # the driver keeps a bounded tail of stdout (BENCH_r04: a 25 KB line did not parse)
    line = json.loads(raw[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "windows", "headline_exactness_verified", "value_f16x3", "detail"):
        assert k in line, k
    assert line["roofline"]["peak_basis"].startswith("fp16 dense") and 0 < line["roofline"]["frac"] < 1
    detail = json.load(open(os.path.join(ROOT, line["detail"])))  # the full object: per-kernel lines, per-mode objects
    assert detail["value"] == pytest.approx(line["value"], rel=1e-6) and detail["roofline"]["kernels"]
    assert line["n_gpus"] == 2 and line["dtype"] == "f16x3" and line["scaling"] == "weak" and line["steps"] == steps
    d = line["dist"]
    assert d["world_size"] == 2 and d["backend"] == "gloo" and d["checksum_equal_on_all_ranks"] is True and d["track_rows"] > 0
    w = line["windows"]
    assert w["n"] == 1 and w["steps_each"] == steps
    wd = detail["windows"]
    assert abs(detail["value"] - world * 64 * steps / (wd["median_ms"] * 1e-3)) <= 1e-6 * detail["value"]  # whole-job frames of the window / its time
    assert abs(detail["ms_per_step"] - wd["median_ms"] / steps) <= 1e-9 * wd["median_ms"]
    assert abs(line["ms_per_step"] - w["median_ms"] / steps) <= 1e-3 * line["ms_per_step"]  # (the line carries rounded figures)
    chk = line["headline_check"]
    assert chk["frames"] == world * 64 and chk["index_mismatches"] == 0 and chk["verified"] is True and line["headline_exactness_verified"] is True
    assert line["config"]["global_batch"] == 128 and "cpu_baseline" not in line and "closed_loop" not in detail


def test_two_rank_pipeline_equals_single_rank(hip_lib, tmp_path):
    import torch

    assert not torch.cuda.is_initialized(), "this test must run before anything touches the GPU in this process"
    world, B, steps = 2, 32, 4
    # three worlds of two ranks, one after the other, all before this process touches the GPU ("hybd": deferred second look — the lanes'
    # all-gathers and ResMLP steps follow their flushes)
    for tag, extra in (("rank", []), ("hyb", ["--hybrid"]), ("hybd", ["--hybrid", "--defer", "2"])):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       HSA_ENABLE_IPC_MODE_LEGACY="0")
            env.pop("GPU_MAX_HW_QUEUES", None)  # the ranks share one device: the runtime default, not eight hardware queues each
            cmd = [sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), "--out", str(tmp_path / f"{tag}{r}.npz"),
                   "--batch", str(B), "--steps", str(steps), "--lanes", "2", "--backend", "gloo"] + extra
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
            assert p.returncode == 0, f"{tag} {r} failed:\n{outs[r][-3000:]}"
        assert not torch.cuda.is_initialized()

    # single-rank run of the same 2*B*steps frames in this process (first GPU use of the parent)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_worker import run_pipeline
    from wtracker_amd import frames as fr

    frames_np, _ = fr.synthetic_frames(steps * B * world, 128, seed=4)
    one = run_pipeline(frames_np, B * world, steps, 0, 1, None, torch.device("cuda", 0), lanes=1)
    assert np.isfinite(one["track"]).any() and one["valid"].sum() >= 10
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        np.testing.assert_array_equal(z["track"], one["track"])  # NaN rows compare equal position-wise
        np.testing.assert_array_equal(z["valid"], one["valid"])
        np.testing.assert_array_equal(z["moves"], one["moves"])
    # the hybrid lanes (fp16 + device-side f16x3 second look, merged BEFORE the exchange): the same rows are replaced however the frames are batched
    hyb = run_pipeline(frames_np, B * world, steps, 0, 1, None, torch.device("cuda", 0), lanes=1, hybrid=True)
    assert int(hyb["replaced"][0]) > 0
    for tag in ("hyb", "hybd"):
        replaced = 0
        for r in range(world):
            z = np.load(tmp_path / f"{tag}{r}.npz")
            np.testing.assert_array_equal(z["track"], hyb["track"], err_msg=tag)
            np.testing.assert_array_equal(z["valid"], hyb["valid"], err_msg=tag)
            np.testing.assert_array_equal(z["moves"], hyb["moves"], err_msg=tag)
            replaced += int(z["replaced"][0])
        assert replaced == int(hyb["replaced"][0]), tag  # every rank replaces the weak rows of its own share


def test_rccl_collectives_in_a_one_rank_world(hip_lib):
    """The N > 1 path's torch.distributed calls against the REAL RCCL on the one visible GPU (world size 1): `device_id=` eager init, barrier,
    all_reduce of CUDA tensors, all_gather_into_tensor into track views from two lane streams — tests/rccl_one_rank.py, a fresh child process.
    (Two ranks on one device are refused by RCCL; two devices are what tests/test_a_gpu_multi_device.py waits for.)"""
    import socket
    import subprocess
    import sys

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_one_rank.py"), str(port)], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "rccl one-rank world ok" in r.stdout, r.stdout[-3000:]
