"""In-tree build of libwtk_hip.so (hipcc, gfx950 only).  Used by __graft_entry__.build()."""
from __future__ import annotations

import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libwtk_hip.so")
SOURCES = ["wtk_api.hip", "wtk_plan.hip", "wtk_run.hip", "wtk_hybrid.hip", "conv_igemm.hip", "conv_sk.hip", "conv1x1_wide.hip", "conv3x3_halo.hip", "conv3x3_c32.hip", "front_fused.hip", "front_fused_split.hip", "c2f_fused.hip", "stem_pool.hip", "head.hip", "mlp.hip", "track_ops.hip", "comm.hip"]
HEADERS = ["wtk_kernels.h", "wtk_internal.h", os.path.join("..", "..", "include", "wtk_hip.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]
STAMP = os.path.join(CSRC, ".flags")


def _flags() -> list:
    """The compile flags of this process: BASE_FLAGS + WTK_EXTRA_HIPCC_FLAGS (diagnostic builds: -DWTK_*_STAMPS interval stamps, results unchanged)."""
    return BASE_FLAGS + os.environ.get("WTK_EXTRA_HIPCC_FLAGS", "").split()


def _stamp_matches() -> bool:
    return os.path.exists(STAMP) and open(STAMP).read() == " ".join(_flags())


def needs_build() -> bool:
    """Stale when the library is missing, older than a source / header, or was built with OTHER flags than this process asks for (the stamp is
    written after a successful link only: a diagnostic library, or objects left behind by a build that failed half way, never pass for current)."""
    if not os.path.exists(LIB_PATH) or not _stamp_matches():
        return True
    t = os.path.getmtime(LIB_PATH)
    for f in SOURCES + HEADERS:
        if os.path.getmtime(os.path.join(CSRC, f)) > t:
            return True
    return False


def source_sha() -> str:
    """sha256[:16] over the kernel sources of libwtk_hip.so (csrc/*.hip + csrc/wtk_kernels.h + csrc/wtk_internal.h; the public C header only declares
    entry points and is left out): the key that ties a committed rocprofv3 artefact under profiles/ to the build it was
    collected on (bench.py only quotes such a file when the key matches)."""
    import hashlib

    h = hashlib.sha256()
    for f in sorted(SOURCES + ["wtk_kernels.h", "wtk_internal.h"]):
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every HIP translation unit for gfx950 and link libwtk_hip.so next to the package."""
    if not force and not needs_build():
        return LIB_PATH
    objs = []
    flags = _flags()
    if any(f.startswith("-DWTK_TIMING") for f in flags) and os.environ.get("WTK_DIAGNOSTIC_BUILD") != "1":
        # timing-ablation macros (tools/patches) make kernels skip work: such a library returns wrong boxes and must never be built by accident
        raise RuntimeError("WTK_EXTRA_HIPCC_FLAGS contains a -DWTK_TIMING* macro (results would be garbage); set WTK_DIAGNOSTIC_BUILD=1 to build it anyway")
    # one object per translation unit, recompiled only when its source, a shared header or the flag set changed.  The stamp is REMOVED before the
    # first object is overwritten and written again only after the link: objects of a build that failed half way (some with the new flags — a
    # -DWTK_*_STAMPS macro changes struct layouts — some with the old) can never be linked together under a stamp that says they match.
    stamp = STAMP
    flag_key = " ".join(flags)
    same_flags = _stamp_matches()
    if os.path.exists(stamp):
        os.remove(stamp)
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, f)) for f in HEADERS)
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        if not force and same_flags and os.path.exists(obj) and os.path.getmtime(obj) > max(hdr_t, os.path.getmtime(os.path.join(CSRC, src))):
            continue
        cmd = [_hipcc(), *flags, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out, file=sys.stderr)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH, *objs, "-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    with open(stamp, "w") as f:
        f.write(flag_key)
    return LIB_PATH


def ensure_built(verbose: bool = False) -> str:
    """Build if stale, serialised over the processes of a node by an exclusive file lock: with N ranks starting together the
    first one through the lock compiles, the others find a fresh library (or hit the same compiler error and exit non-zero too).
    No rendezvous is involved, so a failed build can never leave ranks waiting in a barrier."""
    import fcntl

    if not needs_build():
        return LIB_PATH
    lock = os.path.join(PKG_DIR, ".build.lock")
    with open(lock, "w") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            if needs_build():
                build(force=False, verbose=verbose)
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
