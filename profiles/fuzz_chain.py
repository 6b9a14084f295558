"""Chain mode and mwf_wfa_auto (reference miniwfa.c:850-907) against the COMPILED reference (oracle/_ref/libmwf_ref.so, which travels with
the snapshot): random pairs with structural variation — substitutions / short indels at 1-20 %, long insertions and deletions, tandem
duplications, unrelated blocks of 2-15 kb, low-complexity stretches — over k-mer sizes, occurrence and length filters, score and CIGAR,
low-memory gap fills.  s and the CIGAR must be equal (n_iter too for mwf_wfa_auto).  Usage: python profiles/fuzz_chain.py [seed] [pairs]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import miniwfa_amd as mw
from oracle.pyoracle import Reference, make_opt

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
ref = Reference()
KEYS = ("flag", "x", "o1", "e1", "o2", "e2", "step", "max_s", "max_iter", "max_occ", "kmer", "min_len")

def rnd(n): return rng.integers(0, 4, n).astype(np.uint8)

def make_pair():
    tl = int(rng.choice([300, 1500, 5000, 12000, 30000]))
    t = rnd(tl)
    if rng.random() < 0.3:  # a low-complexity stretch and a tandem repeat in the target
        a = int(rng.integers(0, tl // 2)); t[a:a + tl // 10] = t[a]
        unit = rnd(int(rng.integers(2, 40))); b = int(rng.integers(tl // 2, tl - 1)); n = min(tl - b, len(unit) * 30); t[b:b + n] = np.resize(unit, n)
    q = t.copy()
    p = float(rng.choice([0.01, 0.04, 0.1, 0.2]))
    flip = rng.random(len(q)) < p
    q[flip] = (q[flip] + rng.integers(1, 4, int(flip.sum()))) & 3
    for _ in range(int(rng.integers(0, 4))):  # structural events
        kind = int(rng.integers(0, 4)); at = int(rng.integers(0, max(1, len(q) - 1))); ln = int(rng.choice([50, 400, 2500, 11000]))
        if kind == 0: q = np.concatenate([q[:at], q[at + ln:]])                       # deletion
        elif kind == 1: q = np.concatenate([q[:at], rnd(ln), q[at:]])                 # insertion
        elif kind == 2: q = np.concatenate([q[:at], q[max(0, at - ln):at], q[at:]])   # tandem duplication
        else: q = np.concatenate([q[:at], rnd(ln), q[at + ln:]])                      # a block that does not align
    if len(q) == 0: q = rnd(10)
    return acgt[t].tobytes(), acgt[q].tobytes()

OPTS = [dict(flag=1), dict(flag=0), dict(flag=1, kmer=11, max_occ=3, min_len=20), dict(flag=1, kmer=15, max_occ=1, min_len=40), dict(flag=1, step=200),
        dict(flag=1, kmer=9, max_occ=5, min_len=10), dict(flag=1, x=2, o1=2, e1=2, o2=12, e2=1), dict(flag=1, x=6, o1=2, e1=2, o2=20, e2=1)]
bad = 0
t_ref = t_gpu = 0.0
by_len = {}
for i in range(n_pairs):
    t, q = make_pair()
    for kw in OPTS:
        o = make_opt(**kw)
        t0 = time.time(); es, _, ecig = ref.chain(t, q, o); d_ref = time.time() - t0
        t0 = time.time(); s, _, cig = mw.wfa_chain(t, q, mw.opt_init(**{k: int(getattr(o, k)) for k in KEYS})); d_gpu = time.time() - t0
        t_ref += d_ref; t_gpu += d_gpu
        e = by_len.setdefault(len(t), [0, 0.0, 0.0]); e[0] += 1; e[1] += d_ref; e[2] += d_gpu
        ok = s == es and (None if cig is None else list(cig)) == ecig
        if not ok:
            bad += 1
            if bad <= 5: print("   BAD chain pair", i, len(t), len(q), kw, "got", s, "expected", es, flush=True)
    if len(t) + len(q) <= 30000:  # mwf_wfa_auto: the exact branch below 1e8 cells, the chain beyond
        o = make_opt(flag=1)
        es, eit, ecig = ref.auto(t, q, o)
        s, it, cig = mw.wfa_auto(t, q, mw.opt_init(flag=1))
        if not (s == es and it == eit and (None if cig is None else list(cig)) == ecig):
            bad += 1
            print("   BAD auto pair", i, len(t), len(q), "got", s, it, "expected", es, eit, flush=True)
print(f"seed {seed}: {n_pairs} pairs x {len(OPTS)} option sets, mismatches {bad}; reference {t_ref:.1f} s, this library {t_gpu:.1f} s")
for tl in sorted(by_len): print(f"   target {tl:6d} bp: {by_len[tl][0]:3d} calls, reference {1e3 * by_len[tl][1] / by_len[tl][0]:8.2f} ms per call, this library {1e3 * by_len[tl][2] / by_len[tl][0]:8.2f} ms per call")
print("FUZZ CHAIN", "FAILED" if bad else "OK", "seed", seed)
sys.exit(1 if bad else 0)
