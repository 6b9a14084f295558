#!/usr/bin/env python3
"""Generate tests/golden/*.jsonl by running the REAL reference (oracle/_ref/libmwf_ref.so).

Run in the build container (needs /root/reference to compile the reference):

    python tests/golden/make_golden.py

Every line of the output is one known-answer vector: inputs (literal sequences as hex, or the
(seed, tl, p) of a miniwfa_amd.synth pair), the mwf_opt_t fields, the entry point, and what the
reference returned (s, n_iter, CIGAR string or null).  The fixtures are data only; no reference
source text is stored.
"""
from __future__ import annotations

import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.pyoracle import Reference, make_opt, cigar_str, MWF_F_CIGAR, MWF_F_NO_KALLOC  # noqa: E402
from miniwfa_amd.synth import synth_pair, _stream  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

# the reference's only shipped fixture: test/t3-0.fa, test/t3-1.fa (sequence lines only)
T3_0 = b"CAGGGGCAGACTGACACTTCACACGGCCGGGTACTCTAACAGACCTGCAGCTGAGGGTCCT"
T3_1 = (b"TAGGGGCAGACTGACACCTCACACGGCCGGGTACTCCTCTGAGACAAAACTTCCAGAGGAACGATCAGACAGCAGCATTCGCGGTTCATGAAAATCCGCTGTTCTGCAGCC"
        b"ACCGCTGCTGGTACCCAGGCAAACAGGGTCTAGAGTGGACCTTTAGCAAACTCCAACAGACCTGCAGCTGAGGGTCCT")

DEFAULT = dict(x=4, o1=4, e1=2, o2=15, e2=1)
AFFINE = dict(x=4, o1=4, e1=2, o2=4, e2=2)   # main.c:34  (-a)
EDIT = dict(x=1, o1=0, e1=1, o2=0, e2=1)     # main.c:35  (-e)


def opt_fields(o):
    return {k: getattr(o, k) for k in ("flag", "x", "o1", "e1", "o2", "e2", "step", "max_s", "max_iter")}


def run(R, entry, t, q, o):
    fn = {"exact": R.align, "auto": R.auto, "chain": R.chain}[entry]
    s, n_iter, cig = fn(t, q, o)
    return {"s": s, "n_iter": n_iter, "cigar": None if cig is None else cigar_str(cig)}


def modes(pen, steps=(0, 1, 2, 7, 100), score=True):
    out = []
    if score:
        out.append(make_opt(flag=0, **pen))
    for st in steps:
        out.append(make_opt(flag=MWF_F_CIGAR, step=st, **pen))
    return out


def main():
    R = Reference()
    vec = []

    def lit(name, t, q, o, entry="exact"):
        vec.append({"id": f"{name}#{len(vec)}", "kind": "literal", "t": t.hex(), "q": q.hex(), "entry": entry,
                    "opt": opt_fields(o), "expect": run(R, entry, t, q, o)})

    def syn(name, seed, tl, p, o, entry="exact", n_long=0, long_max=0):
        t, q = synth_pair(seed, tl, p, n_long, long_max)
        vec.append({"id": f"{name}#{len(vec)}", "kind": "synth", "seed": seed, "tl": tl, "p": p, "n_long": n_long,
                    "long_max": long_max, "ql": len(q), "entry": entry, "opt": opt_fields(o),
                    "expect": run(R, entry, t, q, o)})

    # ---- 1. the reference's own fixture under every exact-mode flag combination (main.c:29-44)
    for pen in (DEFAULT, AFFINE, EDIT):
        for o in modes(pen, steps=(0, 1, 3, 5, 50, 5000)):
            lit("t3", T3_0, T3_1, o)
            lit("t3swap", T3_1, T3_0, o)
    lit("t3-K", T3_0, T3_1, make_opt(flag=MWF_F_CIGAR | MWF_F_NO_KALLOC))
    lit("t3-auto", T3_0, T3_1, make_opt(flag=MWF_F_CIGAR), entry="auto")
    lit("t3-auto-score", T3_0, T3_1, make_opt(flag=0), entry="auto")
    lit("t3-chain", T3_0, T3_1, make_opt(flag=MWF_F_CIGAR), entry="chain")

    # ---- 2. tiny hand cases (SURVEY.md Appendix B + a few more)
    hand = [(b"A", b""), (b"", b"ACGTA"), (b"ACGT", b"ACGT"), (b"ACGT", b"acgt"), (b"ACGT", b"AGGT"),
            (b"ACGT", b"ACT"), (b"ACT", b"ACGT"), (b"AAAA", b"TTTT"), (b"ACGTACGT", b"ACGT" + b"T" * 18 + b"ACGT"),
            (b"NNNN", b"NNNN"), (b"GATTACA", b"GCATGCU"), (b"A" * 20, b"A"), (b"A", b"A" * 20), (b"A", b"C"),
            (b"AC", b"CA"), (b"ACGTACGTACGTACGT", b"ACGTACGTTACGTACGT"), (b"A" * 40, b"A" * 17 + b"C" + b"A" * 25),
            (bytes(range(1, 60)), bytes(range(1, 30)) + bytes(range(31, 60))),  # arbitrary byte values
            (bytes([0, 255, 0, 255, 7]), bytes([0, 255, 255, 7])),
            (b"ACGT" * 16, b"ACGT" * 12 + b"TTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTT" + b"ACGT" * 4),
            (b"TTTTTTTTACGTACGTACGTGGGG", b"ACGTACGTACGT")]
    for t, q in hand:
        for pen in (DEFAULT, AFFINE, EDIT):
            for o in modes(pen, steps=(0, 2)):
                if not t and not q and (o.flag & MWF_F_CIGAR):
                    continue
                lit("hand", t, q, o)
    lit("empty-score", b"", b"", make_opt(flag=0))  # CIGAR mode on ("","") crashes the reference (miniwfa.c:406-407)

    # ---- 3. early stops (miniwfa.c:422-428)
    for kw in (dict(max_s=3), dict(max_iter=5), dict(step=2, max_s=3), dict(max_s=15), dict(max_s=16), dict(max_iter=95), dict(max_iter=96)):
        for flag in (0, MWF_F_CIGAR):
            if kw.get("step") and not flag:
                continue
            lit("stop", b"AAAA", b"TTTT", make_opt(flag=flag, **DEFAULT, **kw))
    for kw in (dict(max_s=100), dict(max_iter=5000), dict(max_s=154), dict(max_s=155), dict(max_iter=16874), dict(max_iter=16875)):
        lit("stop-t3", T3_0, T3_1, make_opt(flag=MWF_F_CIGAR, **DEFAULT, **kw))

    # ---- 4. seeded synthetic pairs, default penalties, all modes
    k = 0
    for tl in (1, 2, 7, 31, 64, 100, 257, 300, 1000, 2000):
        for p in (0.0, 0.01, 0.05, 0.15, 0.4):
            seed = 20000 + k
            k += 1
            for o in modes(DEFAULT, steps=(0, 1, 7, 100)):
                t, q = synth_pair(seed, tl, p)
                if not t and not q and (o.flag & MWF_F_CIGAR):
                    continue
                syn("syn", seed, tl, p, o)
    # shrink (every 256 penalties) and phantom offsets matter once s >> 256
    for seed, tl, p in ((30001, 3000, 0.1), (30002, 5000, 0.08), (30003, 4000, 0.25), (30004, 1500, 0.5)):
        for o in modes(DEFAULT, steps=(0, 500, 5000)):
            syn("syn-mid", seed, tl, p, o)
    # unequal lengths / structural differences (long indels push the band against the matrix edge)
    for seed, tl, p, nl, lm in ((31001, 3000, 0.03, 2, 800), (31002, 2000, 0.05, 3, 1500), (31003, 6000, 0.02, 4, 1000)):
        for o in modes(DEFAULT, steps=(0, 256, 1000)):
            syn("syn-sv", seed, tl, p, o, n_long=nl, long_max=lm)

    # ---- 5. random penalty sets
    r = _stream(4242, 9, 400)
    for j in range(40):
        pen = dict(x=1 + int(r[5 * j] % 6), o1=int(r[5 * j + 1] % 7), e1=1 + int(r[5 * j + 2] % 4),
                   o2=int(r[5 * j + 3] % 31), e2=1 + int(r[5 * j + 4] % 3))
        seed = 40000 + j
        tl = (50, 200, 600, 1200)[j % 4]
        p = (0.03, 0.1, 0.25)[j % 3]
        for o in modes(pen, steps=(0, 3, 64)):
            syn("syn-pen", seed, tl, p, o)

    # ---- 6. benchmark-shaped pairs (BASELINE.json configs 3 and 5), score and CIGAR
    for seed in (50000, 50001):
        syn("cfg3", seed, 10000, 0.05, make_opt(flag=0))
        syn("cfg3", seed, 10000, 0.05, make_opt(flag=MWF_F_CIGAR))
    syn("cfg3", 50000, 10000, 0.05, make_opt(flag=MWF_F_CIGAR, step=1000))
    syn("cfg5", 60000, 50000, 0.03, make_opt(flag=0))
    # mwf_wfa_auto: exact branch (n_iter <= 1e8, miniwfa.c:901-903) on a benchmark-shaped pair
    syn("cfg3-auto", 50000, 10000, 0.05, make_opt(flag=MWF_F_CIGAR), entry="auto")
    # chain mode answers, for the "next" row f1; not part of the exact path
    for seed, tl, p in ((70000, 2000, 0.05), (70001, 5000, 0.02), (70002, 20000, 0.03)):
        syn("chain", seed, tl, p, make_opt(flag=MWF_F_CIGAR), entry="chain")
        syn("chain", seed, tl, p, make_opt(flag=0), entry="chain")

    small = [v for v in vec if not v["id"].startswith(("cfg", "chain#")) or v["kind"] == "literal"]
    big = [v for v in vec if v not in small]
    for name, rows in (("exact_small.jsonl", small), ("bench_shaped.jsonl", big)):
        with open(os.path.join(OUT, name), "w") as f:
            for v in rows:
                f.write(json.dumps(v, separators=(",", ":")) + "\n")
        print(name, len(rows), "vectors", os.path.getsize(os.path.join(OUT, name)), "bytes")


if __name__ == "__main__":
    main()
