"""Builds miniwfa_amd/csrc/libmwf_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libmwf_hip.so")
SOURCES = ["mwf_kernels.hip", "mwf_band2.hip", "mwf_lane.hip", "mwf_mid.hip", "mwf_sys.hip", "mwf_engine.cpp", "mwf_memory.cpp", "mwf_plan.cpp", "mwf_chain.cpp", "mwf_async.cpp", "kalloc.cpp", "mwf_dbg.cpp"]
HEADERS = [os.path.join(CSRC, "mwf_internal.h"), os.path.join(CSRC, "mwf_device.h"), os.path.join(CSRC, "mwf_engine.h"), os.path.join(ROOT, "include", "miniwfa.h"), os.path.join(ROOT, "include", "kalloc.h")]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))] + HEADERS + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """One object per source (only stale ones are recompiled, in parallel), then one link."""
    if not force and not stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-I", CSRC]
    hdr_time = max(os.path.getmtime(h) for h in HEADERS + [os.path.abspath(__file__)])
    jobs, objs = [], []
    for name in SOURCES:
        src = os.path.join(CSRC, name)
        if not os.path.exists(src):
            continue
        obj = os.path.join(objdir, name + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            jobs.append([hipcc(), *flags, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        return cmd, subprocess.run(cmd, capture_output=True, text=True)

    with ThreadPoolExecutor(max(1, min(4, len(jobs)))) as ex:
        for cmd, r in ex.map(run, jobs):
            if r.returncode != 0:
                sys.stderr.write(r.stdout + r.stderr)
                raise RuntimeError("hipcc failed: " + " ".join(cmd))
            if verbose and r.stderr.strip():
                sys.stderr.write(r.stderr)
    cmd, r = run([hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-o", LIB + ".tmp", "-lpthread"])
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed linking libmwf_hip.so")
    os.replace(LIB + ".tmp", LIB)
    return LIB


# which source file defines a kernel (by the start of its symbol): what a counter profile of that kernel is a profile OF
KERNEL_FILES = {"wfa_band2_kernel": "mwf_band2.hip", "wfa_sys_kernel": "mwf_sys.hip", "wfa_sys_seg_kernel": "mwf_sys.hip", "wfa_batch_kernel": "mwf_kernels.hip",
                "wfa_lane_kernel": "mwf_lane.hip", "wfa_mid_kernel": "mwf_mid.hip"}


def kernel_fingerprint(symbol: str) -> str | None:
    """SHA-256 (first 16 hex digits) of the source file that defines `symbol` plus the device headers every kernel includes.  profiles/
    make_traffic.py stores it with a kernel's counter traffic; bench.py recomputes it on the tree it runs from and marks `roofline.frac_stale`
    when they differ — the profile then describes other code than the kernel that was timed."""
    import hashlib
    name = symbol.split("<")[0].strip()
    f = KERNEL_FILES.get(name)
    if f is None:
        return None
    h = hashlib.sha256()
    for p in (os.path.join(CSRC, f), os.path.join(CSRC, "mwf_device.h"), os.path.join(CSRC, "mwf_internal.h")):
        try:
            h.update(open(p, "rb").read())
        except OSError:
            return None
    return h.hexdigest()[:16]


CLI = os.path.join(ROOT, "tools", "test-mwf")


def build_cli(force: bool = False) -> str:
    """tools/test-mwf: the reference's command line (main.c) on top of libmwf_hip.so; gzip input when zlib.h is there."""
    src = os.path.join(ROOT, "tools", "test-mwf.cpp")
    if not force and os.path.exists(CLI) and os.path.getmtime(CLI) >= max(os.path.getmtime(src), os.path.getmtime(LIB)):
        return CLI
    zl = ["-DMWF_HAVE_ZLIB", "-lz"] if os.path.exists("/usr/include/zlib.h") else []
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o", CLI,
           "-L", CSRC, "-lmwf_hip", "-Wl,-rpath," + CSRC] + zl
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("g++ failed building tools/test-mwf")
    return CLI


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_cli(force="--force" in sys.argv))
