"""TEST INFRASTRUCTURE (not part of the product package): builds the reference's OWN, unchanged caller (main.c:19-92) against this
repo's headers and libmwf_hip.so, for tests/test_cli.py.  Lives beside oracle/Makefile, which builds the compiled reference; the
binary lands in oracle/_ref/ (git-ignored, travels to the GPU box like the compiled reference)."""
from __future__ import annotations

import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "miniwfa_amd", "csrc")
LIB = os.path.join(CSRC, "libmwf_hip.so")

REF_MAIN = os.path.join(ROOT, "oracle", "_ref", "ref-main-on-libmwf_hip")


def _drop_stale() -> None:
    """A rebuild was due (the sources are here and the binary is older than libmwf_hip.so or main.c) and FAILED: an old binary would be
    linked against last week's library and headers and report the drop-in check as passing — remove it, the test then skips.
    (Where the reference's sources are absent — the GPU box — the binary that travelled with the snapshot is what there is, and is used.)"""
    try:
        os.remove(REF_MAIN)
    except OSError:
        pass
    return None


def build_ref_main(ref: str = "/root/reference") -> str | None:
    """The reference's OWN, unchanged caller (main.c:19-92) compiled against this repo's headers
    and linked with libmwf_hip.so — the drop-in claim with the reference's program rather than ours.  Only where the reference's
    sources are present (the build container); nothing is copied: main.c, ketopt.h and kseq.h are reached through symlinks in a
    scratch directory so that `#include "miniwfa.h"` / "kalloc.h" resolve to include/ instead of the reference's own headers.
    The binary lands in oracle/_ref/ (git-ignored, travels to the GPU box like the compiled reference)."""
    import shutil
    import tempfile
    need = [os.path.join(ref, f) for f in ("main.c", "ketopt.h", "kseq.h")]
    if not all(os.path.exists(f) for f in need) or not os.path.exists("/usr/include/zlib.h") or not os.path.exists(LIB):
        return REF_MAIN if os.path.exists(REF_MAIN) else None
    if os.path.exists(REF_MAIN) and os.path.getmtime(REF_MAIN) >= max(os.path.getmtime(LIB), *(os.path.getmtime(f) for f in need)):
        return REF_MAIN
    os.makedirs(os.path.dirname(REF_MAIN), exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="mwf_refmain_")
    try:
        for f in need:
            os.symlink(f, os.path.join(tmp, os.path.basename(f)))
        cmd = ["gcc", "-O2", "-w", "-I", os.path.join(ROOT, "include"), os.path.join(tmp, "main.c"), "-o", REF_MAIN,
               "-L", CSRC, "-lmwf_hip", "-Wl,-rpath,$ORIGIN/../../miniwfa_amd/csrc", "-lz"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:  # best effort: a test artefact must never break the library's build entry point (tests/test_cli.py skips without it)
            sys.stderr.write(r.stdout + r.stderr)
            sys.stderr.write("warning: could not build the reference's main.c against include/ + libmwf_hip.so; tests/test_cli.py will skip that check\n")
            return _drop_stale()
    except OSError as e:  # no gcc
        sys.stderr.write(f"warning: {e}; tests/test_cli.py will skip the reference-main check\n")
        return _drop_stale()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return REF_MAIN


if __name__ == "__main__":
    print(build_ref_main())
