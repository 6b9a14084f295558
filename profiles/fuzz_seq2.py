"""Fuzz of the 2-bit sequence copy against the byte-wise copy of the packed band kernel:
random lengths (0 ... 3000), low-complexity and repetitive sequences (long exact runs: the per-lane and whole-wave run walkers),
score and CIGAR, the three instantiated penalty sets.  Usage: python profiles/fuzz_seq2.py [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import PackedBatch

mode = "2bit"
seed = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 1
rng = np.random.default_rng(seed)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)

def rand_seq(n, kind):
    if kind == 0: return ACGT[rng.integers(0, 4, n)]
    if kind == 1: return np.full(n, ACGT[rng.integers(0, 4)], dtype=np.uint8)                       # homopolymer
    if kind == 2: u = ACGT[rng.integers(0, 4, rng.integers(1, 40))]; return np.resize(u, n)          # tandem repeat
    return ACGT[rng.choice(4, n, p=[0.85, 0.05, 0.05, 0.05])]                                       # low complexity

def mutate(t, p):
    out = []
    for b in t:
        r = rng.random()
        if r < p / 3: continue
        if r < 2 * p / 3: out.append(ACGT[rng.integers(0, 4)])
        if r < p: out.append(ACGT[rng.integers(0, 4)]); continue
        out.append(b)
    return np.array(out, dtype=np.uint8)

pairs = []
for i in range(600):
    n = int(rng.choice([0, 1, 15, 16, 17, 31, 32, 33, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025])) if i % 3 == 0 else int(rng.integers(0, 3000))
    t = rand_seq(n, i % 4)
    q = mutate(t, float(rng.choice([0.0, 0.01, 0.05, 0.2]))) if i % 7 else rand_seq(int(rng.integers(0, 3000)), (i + 1) % 4)
    pairs.append((t.tobytes(), q.tobytes()))
pk = PackedBatch(pairs)

def run(m, o):
    eng = mw.Engine(0)
    eng.set("seq2bit", 0 if m == "bytes" else 1)
    b = eng.upload(pk); b.align(o); s, it, nc = b.results()
    cig = [b.cigar(i, int(nc[i])).tolist() for i in range(pk.n)] if o.flag else None
    st = eng.stats(); b.free(); eng.close()
    return np.array(s), np.array(it), cig, st.n_retries

bad = 0
for kw in (dict(), dict(flag=1), dict(flag=1, o2=4, e2=2), dict(flag=1, x=1, o1=0, e1=1, o2=0, e2=1), dict(flag=0, x=6, o1=2, e1=2, o2=20, e2=1), dict(flag=1, max_s=300)):
    a = run("bytes", mw.opt_init(**kw)); c = run(mode, mw.opt_init(**kw))
    ok = (a[0] == c[0]).all() and (a[1] == c[1]).all() and a[2] == c[2]
    bad += not ok
    print(f"{kw}: {pk.n} pairs, {mode} identical to bytes: {ok} (retries {a[3]} / {c[3]})", flush=True)
    if not ok:
        d = [i for i in range(pk.n) if a[0][i] != c[0][i] or a[1][i] != c[1][i] or (a[2] and a[2][i] != c[2][i])]
        print("   differing pairs:", d[:10], [(len(pairs[i][0]), len(pairs[i][1]), int(a[0][i]), int(c[0][i])) for i in d[:5]])
sys.exit(1 if bad else 0)
