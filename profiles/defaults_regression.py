"""New defaults (2-bit sequence copies, 16-bit ring rows, idle-slot skip) against the byte-wise / 32-bit paths on one heterogeneous batch:
lengths 30 bp ... 60 kb in every size class, plain ACGT and other alphabets, score and CIGAR.  Usage: python profiles/defaults_regression.py [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(seed)
pairs = []
for i in range(1500):
    tl = int(rng.choice([30, 150, 300, 1000, 3000, 6000, 10000]))
    t, q = synth_pair(800000 + 10000 * seed + i, tl, float(rng.choice([0.0, 0.02, 0.05, 0.12])))
    if i % 37 == 0: t = t.replace(b"A", b"N", 2)
    if i % 53 == 0: t, q = t.lower(), q.lower()
    pairs.append((t, q))
for i in range(300): pairs.append(synth_pair(810000 + 10000 * seed + i, int(rng.integers(12500, 20000)), 0.02))   # generic kernel, 16-bit rows (>= 256 of them)
for i in range(4): pairs.append(synth_pair(820000 + 10000 * seed + i, 60000, 0.01))
pk = PackedBatch(pairs)
def run(new, flag):
    eng = mw.Engine(0)
    if not new: eng.set("seq2bit", 0); eng.set("ring16", 0)
    b = eng.upload(pk); b.align(mw.opt_init(flag=flag)); s, it, nc = b.results()
    cig = [b.cigar(i, int(nc[i])).tolist() for i in range(pk.n)] if flag else None
    st = eng.stats(); b.free(); eng.close()
    return np.array(s), np.array(it), cig, st.n_retries
bad = 0
for flag in (0, 1):
    a = run(False, flag); c = run(True, flag)
    ok = (a[0] == c[0]).all() and (a[1] == c[1]).all() and a[2] == c[2]
    bad += not ok
    print(f"flag={flag}: {pk.n} pairs, new defaults identical to byte-wise / 32-bit paths: {ok} (retries {a[3]} / {c[3]})", flush=True)
sys.exit(1 if bad else 0)
