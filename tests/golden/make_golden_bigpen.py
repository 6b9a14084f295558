#!/usr/bin/env python3
"""Generate tests/golden/big_penalties.jsonl by running the REAL reference (oracle/_ref/libmwf_ref.so) on penalty sets whose ring
is deeper than 256 slices — max(x, o1+e1, o2+e2) >= 256, which the reference accepts like any other (miniwfa.c:390-393) and this
library serves through the generic kernel's big-ring form.  Same vector format as make_golden.py; run in the build container:

    python tests/golden/make_golden_bigpen.py
"""
from __future__ import annotations

import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.pyoracle import Reference, make_opt, MWF_F_CIGAR  # noqa: E402
from miniwfa_amd.synth import synth_pair  # noqa: E402
from make_golden import T3_0, T3_1, opt_fields, run  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

PENS = (dict(x=4, o1=4, e1=2, o2=300, e2=1),       # a long-gap piece that opens at 300
        dict(x=4, o1=254, e1=2, o2=15, e2=1),      # o1+e1 = 256: the first ring depth (257) beyond the fast kernels' tables
        dict(x=4, o1=4, e1=2, o2=254, e2=1),       # o2+e2 = 255: the last depth (256) inside them
        dict(x=300, o1=4, e1=2, o2=15, e2=1),      # a mismatch costs more than any gap the pair needs
        dict(x=6, o1=500, e1=3, o2=1000, e2=1),
        dict(x=2, o1=3000, e1=2, o2=4000, e2=1))   # near the limit (4096)


def main():
    R = Reference()
    vec = []

    def lit(name, t, q, o):
        vec.append({"id": f"{name}#{len(vec)}", "kind": "literal", "t": t.hex(), "q": q.hex(), "entry": "exact",
                    "opt": opt_fields(o), "expect": run(R, "exact", t, q, o)})

    def syn(name, seed, tl, p, o, n_long=0, long_max=0):
        t, q = synth_pair(seed, tl, p, n_long, long_max)
        vec.append({"id": f"{name}#{len(vec)}", "kind": "synth", "seed": seed, "tl": tl, "p": p, "n_long": n_long,
                    "long_max": long_max, "ql": len(q), "entry": "exact", "opt": opt_fields(o), "expect": run(R, "exact", t, q, o)})

    for j, pen in enumerate(PENS):
        modes = [make_opt(flag=0, **pen), make_opt(flag=MWF_F_CIGAR, **pen), make_opt(flag=MWF_F_CIGAR, step=50, **pen),
                 make_opt(flag=MWF_F_CIGAR, step=700, **pen)]
        for o in modes:
            lit("big-t3", T3_0, T3_1, o)
            lit("big-t3swap", T3_1, T3_0, o)
            lit("big-hand", b"ACGTACGT", b"ACGT" + b"T" * 18 + b"ACGT", o)
            lit("big-hand", b"AAAA", b"TTTT", o)
            lit("big-hand", b"", b"ACGTA", o)
            syn("big-syn", 81000 + j, 300, 0.05, o)
            syn("big-syn", 81100 + j, 1000, 0.1, o)
            syn("big-sv", 81200 + j, 1500, 0.03, o, n_long=2, long_max=400)
        syn("big-stop", 81300 + j, 600, 0.1, make_opt(flag=MWF_F_CIGAR, max_s=500, **pen))
        syn("big-stop", 81300 + j, 600, 0.1, make_opt(flag=0, max_iter=200000, **pen))
    with open(os.path.join(OUT, "big_penalties.jsonl"), "w") as f:
        for v in vec:
            f.write(json.dumps(v, separators=(",", ":")) + "\n")
    print("big_penalties.jsonl", len(vec), "vectors")


if __name__ == "__main__":
    main()
