"""Mixed batches under the DEFAULT routing (no forced kernel: what a caller of mwf_wfa_batch gets) against the oracle, score and CIGAR:
thousands of read-length pairs at 0-25 % divergence (lane kernel, device-side re-runs, mid kernel), medium pairs, length-skewed and
unrelated pairs (windows that climb), a few 5-12 kb pairs; four penalty sets (folded and unfolded forms).  s, n_iter and the CIGAR must be equal.
Usage: python profiles/fuzz_default_routing.py [seed] [scale]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import PackedBatch, fuzz_pairs, synth_pair, skewed_pairs
from oracle.pyoracle import Oracle, make_opt

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rng = np.random.default_rng(seed)
orc = Oracle()
pairs = []
for i in range(int(1500 * scale)):
    pairs.append(synth_pair(seed * 100000 + i, int(rng.integers(50, 400)), float(rng.choice([0.0, 0.02, 0.05, 0.1, 0.25]))))
for i in range(int(300 * scale)):
    pairs.append(synth_pair(seed * 100000 + 50000 + i, int(rng.integers(400, 3000)), float(rng.choice([0.01, 0.05, 0.15]))))
pairs += skewed_pairs(seed, int(60 * scale), 200, 3000)
pairs += fuzz_pairs(seed, int(60 * scale), 2500)
for i in range(int(8 * scale)):
    pairs.append(synth_pair(seed * 100000 + 90000 + i, int(rng.integers(5000, 12000)), float(rng.choice([0.03, 0.05, 0.1]))))
order = rng.permutation(len(pairs))
pairs = [pairs[i] for i in order]
pk = PackedBatch(pairs)
bad = 0
for kw in (dict(), dict(x=2, o1=2, e1=2, o2=12, e2=1), dict(x=6, o1=2, e1=2, o2=20, e2=1), dict(x=1, o1=0, e1=1, o2=0, e2=1)):
    t0 = time.time()
    exp = [orc.align(t, q, make_opt(flag=1, **kw)) for t, q in pairs]
    t_or = time.time() - t0
    for flag in (0, 1):
        eng = mw.Engine(0)
        b = eng.upload(pk); b.align(mw.opt_init(flag=flag, **kw)); s, it, nc = b.results()
        n_bad = 0
        for i, (es, eit, ecig) in enumerate(exp):
            ok = (int(s[i]), int(it[i])) == (es, eit) and (not flag or b.cigar(i, int(nc[i])).tolist() == (ecig or []))
            if not ok:
                n_bad += 1
                if n_bad <= 3: print("   BAD pair", i, len(pairs[i][0]), len(pairs[i][1]), "got", int(s[i]), int(it[i]), "expected", es, eit, flush=True)
        bad += n_bad
        print(f"seed {seed} {kw} flag {flag}: {pk.n} pairs, mismatches {n_bad}, re-runs {eng.stats().n_retries} (oracle {t_or:.1f} s)", flush=True)
        b.free(); eng.close()
print("FUZZ DEFAULT ROUTING", "FAILED" if bad else "OK", "seed", seed)
sys.exit(1 if bad else 0)
