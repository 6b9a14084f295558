"""Fuzz of the packed band kernel against the oracle (oracle/mwf_oracle.c, itself pinned to the compiled reference): random lengths
(0 ... 4000, lengths around the 16-base / 64-lane / 256-column granularities), random, homopolymer, tandem-repeat and low-complexity
sequences (long exact runs: the per-lane and whole-wave run walkers), unrelated pairs (wide windows, many shrinks, slots that leave
and re-enter the window), score and CIGAR, the three instantiated penalty sets, forced geometries.  s, n_iter and the CIGAR must be
equal.  Usage: python profiles/fuzz_band2_oracle.py [seed] [pairs]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import PackedBatch, fuzz_pairs
from oracle.pyoracle import Oracle, make_opt

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 240
pairs = fuzz_pairs(seed, n_pairs)
pk = PackedBatch(pairs)
orc = Oracle()
OPT_KEYS = ("flag", "x", "o1", "e1", "o2", "e2", "step", "max_s", "max_iter")
bad = 0
for kw in (dict(), dict(flag=1), dict(flag=1, o2=4, e2=2), dict(flag=1, x=1, o1=0, e1=1, o2=0, e2=1), dict(flag=0, x=6, o1=2, e1=2, o2=20, e2=1)):
    t0 = time.time()
    o = make_opt(**kw)
    exp = [orc.align(t, q, o) for t, q in pairs]
    t_or = time.time() - t0
    for block in (0, 64, 512, 768):
        eng = mw.Engine(0)
        if block: eng.set("force_kind", 2); eng.set("block", block); eng.set("band_pack", 1)
        b = eng.upload(pk); b.align(mw.opt_init(**kw)); s, it, nc = b.results()
        st = eng.stats()
        n_bad = 0
        for i, (es, eit, ecig) in enumerate(exp):
            ok = (int(s[i]), int(it[i])) == (es, eit) and (ecig is None or b.cigar(i, int(nc[i])).tolist() == ecig)
            if not ok:
                n_bad += 1
                if n_bad <= 3: print("   BAD pair", i, len(pairs[i][0]), len(pairs[i][1]), "got", int(s[i]), int(it[i]), "expected", es, eit, flush=True)
        bad += n_bad
        print(f"seed {seed} {kw} block {block or 'auto'}: {pk.n} pairs, mismatches {n_bad}, retries {st.n_retries} (oracle {t_or:.1f} s)", flush=True)
        b.free(); eng.close()
print("FUZZ", "FAILED" if bad else "OK", "seed", seed)
sys.exit(1 if bad else 0)
