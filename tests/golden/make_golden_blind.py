#!/usr/bin/env python3
"""tests/golden/blind_classes.jsonl: known answers of the REAL reference (oracle/_ref/libmwf_ref.so) for the classes of input the
other fixtures are thin on — the ones in which round 5's fuzzers found bit-exactness bugs that no golden vector saw:

  * unrelated pairs of 0.5 - 20 kb (windows that fill the matrix, many shrinks: miniwfa.c:139-171),
  * length-skewed pairs, tl / ql >= 4 either way, related and unrelated (the window's start climbs across chunk boundaries),
  * identical pairs and other long pairs to be run side by side in low-memory mode (checkpoints, miniwfa.c:413-416, 551-601),
  * pairs that sit exactly on the length limits of the host's size classes (mwf_plan.cpp: mwf_gpu_batch_align / choose_kernel).

Run in the build container:   python tests/golden/make_golden_blind.py
Each line: a generator spec (miniwfa_amd.synth.spec_pair — no sequence text), the mwf_opt_t fields, the reference's (s, n_iter, CIGAR
as SHA-256 + op count when long), and `group`: vectors of one group are meant to be aligned in ONE batch by the GPU tests."""
from __future__ import annotations

import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.pyoracle import Reference, make_opt, cigar_str  # noqa: E402
from miniwfa_amd.synth import spec_pair  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "blind_classes.jsonl")
KEYS = ("flag", "x", "o1", "e1", "o2", "e2", "step", "max_s", "max_iter")


def expect(R, t, q, o):
    s, n_iter, cig = R.align(t, q, o)
    e = {"s": s, "n_iter": n_iter, "n_cigar": None if cig is None else len(cig)}
    if cig is None:
        e["cigar"] = None
    elif len(cig) <= 64:
        e["cigar"] = cigar_str(cig)
    else:  # long CIGARs as a digest of the uint32 words (little endian), like long_pairs.jsonl
        import numpy as np
        e["cigar_sha256"] = hashlib.sha256(np.asarray(cig, dtype="<u4").tobytes()).hexdigest()
    return e


def main():
    R = Reference()
    vec = []
    t_all = time.time()

    def add(group, spec, **okw):
        o = make_opt(**okw)
        t, q = spec_pair(spec)
        t0 = time.time()
        vec.append({"id": f"{group}#{len(vec)}", "group": group, "spec": spec, "tl": len(t), "ql": len(q),
                    "opt": {k: getattr(o, k) for k in KEYS}, "expect": expect(R, t, q, o)})
        dt = time.time() - t0
        if dt > 2:
            print(f"   {vec[-1]['id']} {len(t)} x {len(q)}: s {vec[-1]['expect']['s']} n_iter {vec[-1]['expect']['n_iter']:.3g} in {dt:.1f} s", flush=True)

    seed = 710000

    def nxt():
        nonlocal seed
        seed += 10
        return seed

    # ---- 1. unrelated pairs, 0.5 - 20 kb, equal and unequal lengths; score-only and CIGAR; one with non-default penalties
    for tl, ql in ((500, 500), (640, 1000), (1000, 900), (2300, 2300), (2304, 1100), (3100, 4000), (5000, 5000), (5200, 3900),
                   (9000, 9000), (7000, 12000), (14000, 14000), (20000, 20000), (20000, 9000)):
        sp = {"kind": "unrelated", "seed": nxt(), "tl": tl, "ql": ql}
        add("unrelated", sp)
        if tl + ql <= 20000:
            add("unrelated", sp, flag=1)
    add("unrelated", {"kind": "unrelated", "seed": nxt(), "tl": 3000, "ql": 3000}, flag=1, x=2, o1=2, e1=2, o2=12, e2=1)
    add("unrelated", {"kind": "unrelated", "seed": nxt(), "tl": 2500, "ql": 3500}, flag=1, x=6, o1=2, e1=2, o2=20, e2=1)
    add("unrelated", {"kind": "unrelated", "seed": nxt(), "tl": 4000, "ql": 1500}, flag=1, x=1, o1=0, e1=1, o2=0, e2=1)

    # ---- 2. length-skewed pairs, tl / ql >= 4 either way: unrelated (the window's start climbs) and related (query = a mutated piece of the target)
    for big, small in ((1200, 300), (2560, 512), (3000, 300), (5200, 1200), (8000, 1000), (10000, 2400), (16000, 1000), (16000, 4000), (20000, 2500)):
        for swap in (False, True):
            sp = {"kind": "unrelated", "seed": nxt(), "tl": big, "ql": small, "swap": swap}
            add("skewed", sp)
            if big <= 10000:
                add("skewed", sp, flag=1)
            at = (big - small) // 3
            sp = {"kind": "window", "seed": nxt(), "tl": big, "at": at, "w": small, "p": 0.06, "swap": swap}
            add("skewed", sp, flag=1)
            if big <= 8000:
                add("skewed", sp, flag=1, step=200)
    # windows that climb across a 256-column chunk boundary early: a short target against a long unrelated query and vice versa
    for a, b in ((255, 2300), (257, 2300), (300, 3000), (511, 1800), (513, 2600), (700, 2500), (770, 5200), (1023, 4100), (1025, 4100)):
        for swap in (False, True):
            sp = {"kind": "unrelated", "seed": nxt(), "tl": a, "ql": b, "swap": swap}
            add("climb", sp)
            add("climb", sp, flag=1)

    # ---- 3. long pairs to run SIDE BY SIDE in low-memory mode (step > 0), identical pairs among them: two groups of eight
    for g, (step, lens) in enumerate(((97, (2048, 2000, 2100, 2200, 1900, 2300, 2048, 2500)), (500, (12000, 9000, 40000, 10000, 11000, 12000, 15000, 8000)))):
        for j, n in enumerate(lens):
            if j in (0, 6) or (g == 1 and j == 2):
                sp = {"kind": "identical", "seed": nxt(), "tl": n}
            else:
                sp = {"kind": "fit", "seed": nxt(), "tl": n, "ql": n + (j - 3) * 7, "p": (0.1, 0.03)[g]}
            add(f"side-by-side-{g}", sp, flag=1, step=step)

    # ---- 4. pairs exactly on the size classes' length limits (mwf_plan.cpp): tl + ql + 1 == limit - 1, limit, limit + 1
    for limit in (1400, 3600, 8200, 24576):
        for d in (-1, 0, 1):
            total = limit + d - 1               # tl + ql
            tl = total // 2 + 3
            p = 0.05 if limit < 20000 else 0.03
            sp = {"kind": "fit", "seed": nxt(), "tl": tl, "ql": total - tl, "p": p}
            add("class-limit", sp)
            add("class-limit", sp, flag=1)
    # the lane kernel's admission: max(tl, ql) around 400 (320 in small batches), |tl - ql| around 24
    for m in (319, 320, 321, 399, 400, 401):
        for dl in (0, 24, 25):
            sp = {"kind": "fit", "seed": nxt(), "tl": m, "ql": m - dl, "p": 0.05}
            add("class-limit", sp, flag=1)
    # the whole-device kernel's admission for small batches: tl + ql around 15 000 (CIGAR) and 20 000 (score-only)
    for total, flag in ((14999, 1), (15000, 1), (15001, 1), (19999, 0), (20000, 0), (20001, 0)):
        tl = total // 2
        add("class-limit", {"kind": "fit", "seed": nxt(), "tl": tl, "ql": total - tl, "p": 0.04}, flag=flag)
    # 16-bit offsets: target length + penalty bound around 32 767 (the packed kernels' range rule)
    for tl in (10600, 10700, 10800):
        add("class-limit", {"kind": "fit", "seed": nxt(), "tl": tl, "ql": tl - 40, "p": 0.03})

    with open(OUT, "w") as f:
        for v in vec:
            f.write(json.dumps(v, separators=(",", ":")) + "\n")
    groups = {}
    for v in vec:
        groups[v["group"]] = groups.get(v["group"], 0) + 1
    print(os.path.basename(OUT), len(vec), "vectors", os.path.getsize(OUT), "bytes", groups, f"{time.time() - t_all:.0f} s")


if __name__ == "__main__":
    main()
