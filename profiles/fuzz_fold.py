"""Folded against unfolded form of the packed band kernel (mwf_band2.hip: FOLD) on random batches, score-only: s and n_iter must be equal.
Shapes that move the window's start up (length-skewed and unrelated pairs, long gaps), fuzz pairs, 10-20 kb pairs; penalty sets with o1 == x.
Usage: python profiles/fuzz_fold.py [seed] [pairs]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import PackedBatch, fuzz_pairs, synth_pair, skewed_pairs
from oracle.pyoracle import Oracle, make_opt
orc = Oracle()

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
sets = {
    "skewed 200-3000": skewed_pairs(seed, n, 200, 3000),
    "fuzz": fuzz_pairs(seed, n, 3000),
    "skewed 4-9 kb": skewed_pairs(seed + 7, max(n // 8, 16), 4000, 9000),
    "10 kb @ 5 %": [synth_pair(seed * 1000 + i, 10000, 0.05) for i in range(64)],
    "16 kb @ 4 %": [synth_pair(seed * 1000 + 500 + i, 16000, 0.04) for i in range(32)],
}
bad = 0
for name, pairs in sets.items():
    pk = PackedBatch(pairs)
    for kw in (dict(), dict(x=2, o1=2, e1=2, o2=12, e2=1), dict(x=6, o1=6, e1=1, o2=30, e2=1), dict(x=3, o1=3, e1=2, o2=9, e2=2)):
        if max(len(t) + len(q) for t, q in pairs) < 7000:  # the oracle finishes these in seconds: score-only and CIGAR under the default routing
            o = make_opt(flag=1, **kw)
            exp = [orc.align(t, q, o) for t, q in pairs]
            for flag in (0, 1):
                eng = mw.Engine(0)
                b = eng.upload(pk); b.align(mw.opt_init(flag=flag, **kw)); s, it, nc = b.results()
                n_bad = sum(1 for i, (es, eit, ecig) in enumerate(exp) if (int(s[i]), int(it[i])) != (es, eit) or (flag and b.cigar(i, int(nc[i])).tolist() != (ecig or [])))
                bad += n_bad
                print(f"seed {seed} {name} {kw} flag {flag} default routing against the oracle: mismatches {n_bad}", flush=True)
                b.free(); eng.close()
        for block in (0, 512, 1024):
            res = {}
            for fold in (1, 0):
                eng = mw.Engine(0); eng.set("band_fold", fold)
                if block == 1024: eng.set("band_span", 2)   # (the span geometry: 1024 threads x 5 slots)
                elif block: eng.set("force_kind", 2); eng.set("block", block); eng.set("band_pack", 1)
                b = eng.upload(pk); b.align(mw.opt_init(**kw)); s, it, _ = b.results()
                res[fold] = (s.copy(), it.copy(), eng.stats().n_retries)
                b.free(); eng.close()
            diff = np.nonzero((res[0][0] != res[1][0]) | (res[0][1] != res[1][1]))[0]
            bad += len(diff)
            print(f"seed {seed} {name} {kw} block {block or 'auto'}: {pk.n} pairs, differ {len(diff)} {diff[:5].tolist()}, retries folded {res[1][2]} unfolded {res[0][2]}", flush=True)
print("FUZZ FOLD", "FAILED" if bad else "OK", "seed", seed)
sys.exit(1 if bad else 0)
