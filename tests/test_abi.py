"""CPU-only checks of the drop-in boundary: libwtk_hip.so builds, loads, exports every symbol that
include/wtk_hip.h declares, and its own conv table equals the Python-side spec.  No compute calls."""
import ctypes
import os
import re

import pytest

from wtracker_amd import hip
from wtracker_amd import yolo_spec as ys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "wtk_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wtk_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_list_agree():
    assert header_symbols() == sorted(hip.SYMBOLS)


def test_library_exports_every_declared_symbol(hip_lib):
    for name in header_symbols():
        assert hasattr(hip_lib, name), name
    assert hip_lib.wtk_abi_version() == 7
    assert hip_lib.wtk_last_error() is not None
    assert hip_lib.wtk_device_count() >= 0


def test_binding_argument_counts_match_the_header(hip_lib):
    """Every prototype of include/wtk_hip.h against the ctypes `argtypes` wtracker_amd.hip.load() declares: a parameter added to one side only
    would otherwise surface as a corrupted call on the GPU box."""
    text = open(os.path.join(ROOT, "include", "wtk_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = re.findall(r"\b(wtk_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S)
    assert sorted({n for n, _ in protos}) == header_symbols()
    for name, params in protos:
        p = params.strip()
        n = 0 if p in ("", "void") else p.count(",") + 1
        at = getattr(hip_lib, name).argtypes
        assert (0 if at is None else len(at)) == n, (name, n, at)


@pytest.mark.parametrize("scale,nc", [("n", 1), ("s", 1), ("s", 80), ("m", 3)])
def test_conv_table_matches_python_spec(hip_lib, scale, nc):
    w, d, m = ys.scale_params(scale)
    assert hip.yolo_conv_table(w, d, m, nc) == ys.conv_table(scale, nc)


def test_bad_scale_is_reported(hip_lib):
    assert hip_lib.wtk_yolo_conv_count(ctypes.c_float(0.0), ctypes.c_float(0.33), 1024, 1) == -1
    with pytest.raises(hip.WtkError):
        hip.yolo_conv_table(0.5, 0.33, 1024, 0)


def test_product_path_fails_loudly_without_gpu(hip_lib):
    """No CPU fallback: with no visible device every create call must raise, never compute."""
    if hip.device_count() > 0:
        pytest.skip("a GPU is visible")
    import numpy as np

    with pytest.raises(hip.WtkError, match="GPU"):
        hip.HipMLP([(np.zeros((2, 28), np.float32), np.zeros(2, np.float32), False)] * 2, 0, 0)
    with pytest.raises(hip.WtkError, match="GPU"):
        hip.HipYolo(ys.synthetic_weights("n", 1), (64, 64), 1, width=0.25)


def test_communicator_fails_loudly_without_gpu(hip_lib):
    if hip.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(hip.WtkError, match="GPU"):
        hip.WtkComm(0, 0, 1, bytes(hip.COMM_ID_BYTES))
    with pytest.raises(hip.WtkError, match="128"):
        hip.WtkComm(0, 0, 1, b"short")


def test_no_product_module_imports_the_oracle():
    pkg = os.path.join(ROOT, "wtracker_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_one_hip_runtime_whatever_the_import_order(hip_lib):
    """libwtk_hip.so loaded BEFORE torch must not bring a second HIP / HSA runtime into the process (the one that initialises second
    would see no GPU): hip.load() maps torch's copies first, so both orders end with exactly one libamdhip64 and one libhsa-runtime64."""
    import subprocess
    import sys

    probe = ("import sys; sys.path.insert(0, %r); %s; "
             "print(sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'amdhip64' in l or 'hsa-runtime64' in l)))")
    for order in ("from wtracker_amd import hip; hip.load(); import torch", "import torch; from wtracker_amd import hip; hip.load()"):
        out = subprocess.run([sys.executable, "-c", probe % (ROOT, order)], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        libs = eval(out.stdout.strip().splitlines()[-1])
        assert len([p for p in libs if "amdhip64" in p]) == 1 and len([p for p in libs if "hsa-runtime64" in p]) == 1, (order, libs)


def test_load_leaves_the_environment_alone(hip_lib):
    """hip.load() used to write GPU_MAX_HW_QUEUES=8 into os.environ, which every child process inherits (rank processes sharing one
    device then asked for eight hardware queues each).  The request is now the explicit hip.request_hw_queues(): after load() neither this process
    nor a child spawned from it sees the variable unless somebody asked for it."""
    import subprocess
    import sys

    code = ("import os, sys; sys.path.insert(0, %r); os.environ.pop('GPU_MAX_HW_QUEUES', None); from wtracker_amd import hip; hip.load(); "
            "assert 'GPU_MAX_HW_QUEUES' not in os.environ; "
            "import subprocess; print(subprocess.run([sys.executable, '-c', \"import os; print(os.environ.get('GPU_MAX_HW_QUEUES'))\"], capture_output=True, text=True).stdout.strip()); "
            "assert hip.request_hw_queues(8) and os.environ['GPU_MAX_HW_QUEUES'] == '8' and not hip.request_hw_queues(5) and os.environ['GPU_MAX_HW_QUEUES'] == '8'") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == "None"
