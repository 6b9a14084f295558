#!/usr/bin/env python3
"""Generate tests/golden/chain_fresh.jsonl by running the REAL reference's chain mode (oracle/_ref/libmwf_ref.so:
mwf_wfa_chain, mwf_wfa_auto; reference miniwfa.c:850-907).

    python tests/golden/make_golden_chain.py

Row f1 of SURVEY.md section 8 used to be pinned by seven stored vectors; the pairs of
tests/test_gpu_parity.py::test_chain_and_auto_against_reference were only compared where the compiled reference travelled
with the snapshot.  Their answers are stored here, plus: a pair whose exact alignment needs more than 1e8 cells (mwf_wfa_auto
falls through to the chain), a pair with a >= 10 kb block that does not align at all (miniwfa.c:869, the `mwf_ksim < 0.02`
branch: one deletion + one insertion, no gap fill), k-mer sizes 11 and 15, and a run without kalloc.  Data only: seeds,
options, and what the reference returned."""
from __future__ import annotations

import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.pyoracle import Reference, make_opt, cigar_str  # noqa: E402
from miniwfa_amd.synth import synth_pair, synth_diverged_block  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "chain_fresh.jsonl")
KEYS = ("flag", "x", "o1", "e1", "o2", "e2", "step", "max_s", "max_iter", "max_occ", "kmer", "min_len")


def main():
    R = Reference()
    rows = []

    def add(name, entry, gen, t, q, kw):
        o = make_opt(**kw)
        s, n_iter, cig = (R.chain if entry == "chain" else R.auto)(t, q, o)
        rows.append({"id": f"{name}#{len(rows)}", "entry": entry, "gen": gen, "tl": len(t), "ql": len(q),
                     "opt": {k: int(getattr(o, k)) for k in KEYS},
                     "expect": {"s": s, "n_iter": n_iter if entry == "auto" else None, "cigar": None if cig is None else cigar_str(cig)}})

    for j in range(10):
        args = (89000 + j, (400, 3000, 12000, 30000)[j % 4], (0.02, 0.06, 0.15)[j % 3], j % 3, 700)
        t, q = synth_pair(*args)
        for kw in (dict(flag=1), dict(flag=0), dict(flag=1, kmer=11, max_occ=3, min_len=20), dict(flag=1, step=200)):
            add("fresh", "chain", {"kind": "synth", "args": list(args)}, t, q, kw)
    # mwf_wfa_auto: falls through to the chain beyond 1e8 cells (miniwfa.c:901-907); stays exact below
    for args in ((89100, 60000, 0.06, 0, 0), (89101, 8000, 0.03, 0, 0)):
        t, q = synth_pair(*args)
        add("auto", "auto", {"kind": "synth", "args": list(args)}, t, q, dict(flag=1))
    t, q = synth_pair(89100, 60000, 0.06)
    add("auto-score", "auto", {"kind": "synth", "args": [89100, 60000, 0.06, 0, 0]}, t, q, dict(flag=0))
    # a block of >= 10 kb on both sequences that does not align: bridged by 1 D + 1 I (miniwfa.c:869); and one just below 10 kb (gap fill)
    for args in ((89200, 3000, 12000, 11000, 0.02), (89210, 2000, 10500, 15000, 0.04), (89220, 3000, 9000, 12000, 0.02)):
        t, q = synth_diverged_block(*args)
        for kw in (dict(flag=1), dict(flag=0), dict(flag=1, kmer=15)):
            add("diverged", "chain", {"kind": "diverged", "args": list(args)}, t, q, kw)
    # k-mer size 15 (and 11 with tight occurrence / length filters) on ordinary pairs
    for j, args in enumerate(((89300, 5000, 0.03, 0, 0), (89301, 20000, 0.05, 2, 900), (89302, 1500, 0.1, 0, 0))):
        t, q = synth_pair(*args)
        for kw in (dict(flag=1, kmer=15), dict(flag=1, kmer=15, max_occ=1, min_len=40), dict(flag=1, kmer=9, max_occ=5, min_len=10)):
            add("kmer", "chain", {"kind": "synth", "args": list(args)}, t, q, kw)
    with open(OUT, "w") as f:
        for v in rows:
            f.write(json.dumps(v, separators=(",", ":")) + "\n")
    print(OUT, len(rows), "vectors", os.path.getsize(OUT), "bytes")
    n_bridge = sum(1 for v in rows if v["id"].startswith("diverged") and v["expect"]["cigar"] and any(tok for tok in [v["expect"]["cigar"]] if "D" in tok))
    print("diverged rows:", [(v["id"], v["expect"]["s"]) for v in rows if v["id"].startswith("diverged")], n_bridge)


if __name__ == "__main__":
    main()
