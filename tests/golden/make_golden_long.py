#!/usr/bin/env python3
"""Generate tests/golden/long_pairs.jsonl: the single-pair configurations of BASELINE.json (configs[1]: a C4-like
150 kb pair, configs[3]: an MHC-like 5 Mb pair) run through the REAL reference (oracle/_ref/libmwf_ref.so).

Run in the build container (needs /root/reference to compile the reference); minutes of CPU for the 5 Mb pair:

    python tests/golden/make_golden_long.py            # every vector, one process per vector
    python tests/golden/make_golden_long.py mhc-lowmem # a single vector (prints its line)

The CIGARs of these pairs have tens of thousands of operations, so a vector stores n_cigar and the SHA-256 of the
CIGAR as little-endian uint32 words (len<<4|op, miniwfa.h:50) instead of the string.  Data only.
"""
from __future__ import annotations

import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "long_pairs.jsonl")

# name -> (seed, tl, p, n_long, long_max, opt fields).  Generators: miniwfa_amd.synth.synth_pair (SURVEY.md §8d stand-ins)
C4 = (2001, 150000, 0.035, 0, 0)
MHC = (2002, 5000000, 0.008, 3, 15000)
VECTORS = {
    "c4-score": (*C4, dict(flag=0)),
    "c4-cigar": (*C4, dict(flag=1)),
    "c4-lowmem": (*C4, dict(flag=1, step=5000)),
    "c4-lowmem1000": (*C4, dict(flag=1, step=1000)),
    "mhc-score": (*MHC, dict(flag=0)),               # n_iter of the high-memory core pass (score and CIGAR modes count alike)
    "mhc-lowmem": (*MHC, dict(flag=1, step=5000)),   # BASELINE configs[3]
}


def cigar_sha256(words) -> str:
    import numpy as np
    return hashlib.sha256(np.asarray(words, dtype="<u4").tobytes()).hexdigest()


def one(name: str) -> dict:
    from oracle.pyoracle import Reference, make_opt
    from miniwfa_amd.synth import synth_pair
    seed, tl, p, n_long, long_max, kw = VECTORS[name]
    t, q = synth_pair(seed, tl, p, n_long, long_max)
    o = make_opt(**kw)
    R = Reference(arena=True)
    t0 = time.perf_counter()
    s, n_iter, cig = R.align(t, q, o)
    dt = time.perf_counter() - t0
    return {"id": name, "kind": "synth", "seed": seed, "tl": tl, "p": p, "n_long": n_long, "long_max": long_max, "ql": len(q),
            "entry": "exact",
            "opt": {k: getattr(o, k) for k in ("flag", "x", "o1", "e1", "o2", "e2", "step", "max_s", "max_iter")},
            "expect": {"s": s, "n_iter": n_iter, "n_cigar": None if cig is None else len(cig),
                       "cigar_sha256": None if cig is None else cigar_sha256(cig)},
            "reference_wall_s": round(dt, 2), "reference_host": "build container, 1 thread, gcc -O3 -msse4.2, one kalloc arena"}


def main():
    if len(sys.argv) > 1:
        print(json.dumps(one(sys.argv[1]), separators=(",", ":")))
        return
    procs = {n: subprocess.Popen([sys.executable, os.path.abspath(__file__), n], stdout=subprocess.PIPE, text=True) for n in VECTORS}
    rows = []
    for n, pr in procs.items():
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise SystemExit(f"{n} failed")
        rows.append(out.strip().splitlines()[-1])
        print(rows[-1], flush=True)
    with open(OUT, "w") as f:
        f.write("\n".join(rows) + "\n")
    print(OUT, len(rows), "vectors")


if __name__ == "__main__":
    main()
