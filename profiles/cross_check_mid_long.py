"""Cross-check of the default routing (512-thread geometries on biased offsets, span geometry, re-runs) against the generic kernel on random pairs of 9-24 kb at
divergences 0-12 %: s, n_iter and every CIGAR word must agree.  Usage: python profiles/cross_check_mid_long.py [n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(2024)
pairs = [synth_pair(int(rng.integers(1 << 30)), int(rng.integers(9000, 24000)), float(rng.choice([0.0, 0.01, 0.03, 0.05, 0.08, 0.12])), int(rng.integers(0, 3)), 1500) for _ in range(n)]
pk = PackedBatch(pairs)
bad = 0
for kw in (dict(), dict(flag=1)):
    out = []
    for generic in (1, 0):
        eng = mw.Engine(0)
        if generic: eng.set("force_kind", 0)
        b = eng.upload(pk); b.align(mw.opt_init(**kw)); s, it, nc = b.results()
        cig = [b.cigar(i, int(nc[i])).tolist() for i in range(n)] if kw else None
        out.append((np.array(s), np.array(it), cig, eng.stats().n_retries))
        b.free(); eng.close()
    same = (out[0][0] == out[1][0]).all() and (out[0][1] == out[1][1]).all() and out[0][2] == out[1][2]
    bad += not same
    print(f"{kw}: {n} pairs, default routing re-runs {out[1][3]}, equal to the generic kernel: {bool(same)}", flush=True)
sys.exit(1 if bad else 0)
