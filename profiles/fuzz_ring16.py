"""Fuzz of the generic kernel's 16-bit ring rows (ring16 = 2) against 32-bit rows: random lengths 4 ... 40 kb (the batch's longest pair decides
that the E2/F2-in-LDS path runs), unequal lengths, random and low-complexity sequences, divergence 0 ... 15 %, score and CIGAR, two penalty sets with
e2 = 1.  Usage: python profiles/fuzz_ring16.py [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(seed)
pairs = []
for i in range(40):
    tl = int(rng.integers(4000, 40000))
    t, q = synth_pair(900000 + 1000 * seed + i, tl, float(rng.choice([0.0, 0.005, 0.02, 0.06, 0.15])))
    if i % 5 == 0: q = q[: max(1, len(q) - int(rng.integers(0, 3000)))]          # much shorter query
    if i % 7 == 0: t = t[:1000] + t[1000:2000] * 3 + t[2000:]                   # a tandem repeat in the target
    if i % 11 == 0: t = b"A" * 5000 + t; q = b"A" * 4990 + q                    # a long homopolymer run
    pairs.append((t, q))
pk = PackedBatch(pairs)

def run(r16, o):
    eng = mw.Engine(0); eng.set("ring16", r16); eng.set("force_kind", 0)
    b = eng.upload(pk); b.align(o); s, it, nc = b.results()
    cig = [b.cigar(i, int(nc[i])).tolist() for i in range(pk.n)] if o.flag else None
    st = eng.stats(); b.free(); eng.close()
    return np.array(s), np.array(it), cig, st.n_retries, st.packed

bad = 0
for kw in (dict(), dict(flag=1), dict(flag=1, x=6, o1=2, e1=2, o2=20, e2=1), dict(flag=0, max_s=2000)):
    a = run(0, mw.opt_init(**kw)); c = run(2, mw.opt_init(**kw))
    ok = (a[0] == c[0]).all() and (a[1] == c[1]).all() and a[2] == c[2]
    bad += not ok
    print(f"{kw}: {pk.n} pairs, 16-bit rows (stats.packed {c[4]}) identical to 32-bit: {ok} (retries {a[3]} / {c[3]}), max s {int(a[0].max())}", flush=True)
    if not ok:
        d = [i for i in range(pk.n) if a[0][i] != c[0][i] or a[1][i] != c[1][i] or (a[2] and a[2][i] != c[2][i])]
        print("   differing pairs:", [(i, len(pairs[i][0]), len(pairs[i][1]), int(a[0][i]), int(c[0][i]), int(a[1][i]), int(c[1][i])) for i in d[:6]])
sys.exit(1 if bad else 0)
