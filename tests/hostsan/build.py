"""TEST INFRASTRUCTURE: builds tests/hostsan/_build/hostsan_driver — the HOST side of every translation unit of libwtk_hip.so (hipcc --cuda-host-only:
no device code is generated, kernels become launch stubs) with -fsanitize=address,undefined, linked against hip_stub.cpp (the no-op launch layer that
models device memory and the stream / event / capture protocol) and driver.cpp (the shape matrix).  `python tests/hostsan/build.py [quick|full]` builds
and runs it; tests/test_host_sanitizer.py does the same inside the CPU suite."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "wtracker_amd", "csrc")
OUT = os.path.join(HERE, "_build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CLANGXX = "/opt/rocm/lib/llvm/bin/clang++"
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
HOST_ONLY = ["--cuda-host-only", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]


def _sources() -> list:
    sys.path.insert(0, ROOT)
    from wtracker_amd import _build

    return list(_build.SOURCES)


def _newer(target: str, deps: list) -> bool:
    return os.path.exists(target) and all(os.path.getmtime(target) > os.path.getmtime(d) for d in deps)


def _run(cmd: list) -> None:
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: " + " ".join(cmd) + "\n" + r.stdout)


def build(verbose: bool = False) -> str:
    os.makedirs(OUT, exist_ok=True)
    headers = [os.path.join(CSRC, "wtk_kernels.h"), os.path.join(CSRC, "wtk_internal.h"), os.path.join(ROOT, "include", "wtk_hip.h")]
    jobs, objs = [], []
    for src in _sources():
        obj = os.path.join(OUT, src.replace(".hip", ".o"))
        objs.append(obj)
        if not _newer(obj, [os.path.join(CSRC, src)] + headers):
            jobs.append([HIPCC, *HOST_ONLY, *SAN, "-c", os.path.join(CSRC, src), "-o", obj])
    stub = os.path.join(OUT, "hip_stub.o")
    if not _newer(stub, [os.path.join(HERE, "hip_stub.cpp"), os.path.join(HERE, "launch_checks.inc")] + headers):
        jobs.append([HIPCC, "-x", "hip", *HOST_ONLY, *SAN, "-c", os.path.join(HERE, "hip_stub.cpp"), "-o", stub])
    drv = os.path.join(OUT, "driver.o")
    if not _newer(drv, [os.path.join(HERE, "driver.cpp")] + headers):
        jobs.append([CLANGXX, "-std=c++17", "-Wall", *SAN, "-c", os.path.join(HERE, "driver.cpp"), "-o", drv])
    if verbose:
        for j in jobs:
            print(" ".join(j), file=sys.stderr)
    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(_run, jobs))
    # every host object refers to the device image hipcc would have embedded (__hip_fatbin_<hash>): give each name one dummy word
    names = set()
    for o in objs:
        for line in subprocess.run(["nm", "-u", o], stdout=subprocess.PIPE, text=True).stdout.splitlines():
            sym = line.split()[-1]
            if sym.startswith("__hip_fatbin"):
                names.add(sym)
    fat = os.path.join(OUT, "fatbin_syms.c")
    text = "".join(f"const unsigned long long {n}[4] = {{0}};\n" for n in sorted(names))
    if not os.path.exists(fat) or open(fat).read() != text:
        with open(fat, "w") as f:
            f.write(text)
    exe = os.path.join(OUT, "hostsan_driver")
    if jobs or not os.path.exists(exe):
        _run([CLANGXX, *SAN, "-o", exe, *objs, stub, drv, "-x", "c", fat, "-x", "none", "-ldl", "-lpthread"])
    return exe


def run(mode: str = "quick", env_extra: dict | None = None, timeout: int = 900) -> subprocess.CompletedProcess:
    exe = build()
    env = dict(os.environ)
    for k in [k for k in env if k.startswith("WTK_") or k == "GPU_MAX_HW_QUEUES"]:
        del env[k]
    env["ASAN_OPTIONS"] = "detect_leaks=1:abort_on_error=0:halt_on_error=1:detect_stack_use_after_return=1:strict_string_checks=1"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    env["LSAN_OPTIONS"] = "suppressions=" + os.path.join(HERE, "lsan.supp")
    env.update(env_extra or {})
    return subprocess.run([exe, mode], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=timeout)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "quick"
    extra = dict(a.split("=", 1) for a in sys.argv[2:])
    r = run(mode, extra)
    print(r.stdout)
    sys.exit(r.returncode)
