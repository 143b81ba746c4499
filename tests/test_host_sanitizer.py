"""The HOST side of libwtk_hip.so under AddressSanitizer + UndefinedBehaviorSanitizer, in the build container (no GPU).

tests/hostsan/build.py compiles every translation unit with `hipcc --cuda-host-only -fsanitize=address,undefined` (no device code; kernels become launch
stubs) and links it against tests/hostsan/hip_stub.cpp — a no-op launch layer that models device memory, streams, events and stream capture, and checks
every launch's argument struct against the live allocations — and tests/hostsan/driver.cpp, which drives the C ABI over the GPU suite's shape matrix
(scales n / s / m, 32 x 32 ... 640 x 640 here, 1280 x 1280 and B = 256 in `full` mode; every dtype, both plans, every entry point, handles created and
destroyed out of phase, two host threads).  What it pins: planning, weight packing, launcher geometry, allocation sizes against kernel extents, and the
stream / event / graph lifetime protocol of csrc/wtk_plan.hip / wtk_run.hip (the call of yolo_controller.py:64-90, 96-98 that must never die)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsan"))
import build as hostsan  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(hostsan.HIPCC), reason="hipcc not installed")


def _ok(r):
    tail = "\n".join(r.stdout.splitlines()[-60:])
    assert r.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in r.stdout and "runtime error:" not in r.stdout and "VIOLATION" not in r.stdout, tail


def test_launch_layer_flags_deliberate_violations():
    """The checker checks: eleven protocol / memory violations committed on purpose through the stub's HIP API are all reported."""
    r = hostsan.run("selftest")
    assert r.returncode == 0 and "11 of 11 deliberate violations flagged" in r.stdout, r.stdout[-3000:]


@pytest.mark.parametrize("env", [
    {},  # a controller user's environment: eager launches, latency plan for small handles, launch choices timed at the first call of a batch size (the stub's
    #      timer returns a constant: every candidate of every launch is still launched and checked)
    {"WTK_GRAPH": "1", "WTK_SK_AUTOTUNE": "0"},  # replayed captures on (host entry points and caller buffers), forked through the shared side streams
    # the GPU suite's environment (tests/conftest.py) + captures: the configuration of both round-5 crashes
    {"WTK_LATENCY_PLAN": "0", "WTK_NO_SK_MIXED": "1", "WTK_SMALL_NARROW": "0", "WTK_GRAPH": "1", "WTK_SK_AUTOTUNE": "0"},
    {"WTK_SK_GROUP": "0", "WTK_SK_AUTOTUNE": "0"},  # the latency plan as one launch per conv (the bit-identity reference of the grouped launches)
], ids=["default", "graphs", "suite-env-graphs", "ungrouped"])
def test_host_side_is_clean_over_the_shape_matrix(env):
    r = hostsan.run("quick", env)
    _ok(r)
    assert "violations 0; driver failures 0" in r.stdout, r.stdout[-3000:]


def test_host_side_is_clean_over_the_full_matrix():
    """`full` mode: 120 handles up to 1280 x 1280 and B = 256 (large device regions are address-space reservations: nothing is touched), ~50 s."""
    r = hostsan.run("full", {"WTK_SK_AUTOTUNE": "0"}, timeout=1500)
    _ok(r)
    assert "violations 0; driver failures 0" in r.stdout, r.stdout[-3000:]
