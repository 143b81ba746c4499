import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "tests") not in sys.path:  # `harness` (tests/harness: the stand-in driver) and `dist_worker`
    sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The suites written before round 5 build small handles (max_batch <= 16) to exercise the THROUGHPUT kernels (window / implicit-GEMM kernels, fused
# tails, their bit-identity switches).  Since round 5 wtk_yolo_create plans such handles for latency (split-K convs, conv_sk.hip); the variable keeps
# those suites on the kernels they were written for.  tests/test_gpu_latency.py asks for the latency plan by name (an explicit plan beats the
# variable) and checks the AUTO rule with the variable removed.
os.environ.setdefault("WTK_LATENCY_PLAN", "0")
# ... and (same reason) off the split-K kernel that small throughput-plan handles (max_batch <= 16) use for their smallest maps; test_gpu_latency.py
# removes the variable where it tests that rule
os.environ.setdefault("WTK_NO_SK_MIXED", "1")
os.environ.setdefault("WTK_SMALL_NARROW", "0")
# a SIGSEGV inside the library or the HIP runtime prints its native frames before Python's faulthandler prints the Python stack (csrc/wtk_plan.hip)
os.environ.setdefault("WTK_SEGV_BACKTRACE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """test_a_gpu_*.py start rank processes and must do so before this process has touched the GPU: keep them first
    (the RCCL multi-device module before the one-device gloo rehearsal, whose comparison run initialises the GPU here)."""
    items.sort(key=lambda it: 0 if "test_a_gpu_multi_device" in it.nodeid else 1 if "test_a_gpu_two_ranks" in it.nodeid else 2)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def hip_lib():
    """libwtk_hip.so, built in-tree if missing (hipcc cross-compiles without a GPU)."""
    from wtracker_amd import _build, hip

    if _build.needs_build():
        _build.build(verbose=False)
    return hip.load()
