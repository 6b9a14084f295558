import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
from oracle.pyoracle import Oracle, make_opt
orc = Oracle()
for tl, p, n in ((300, 0.05, 1), (2000, 0.05, 1), (2000, 0.05, 4), (10000, 0.05, 2)):
    pairs = [synth_pair(50000 + i, tl, p) for i in range(n)]
    eng = mw.Engine(0); eng.set("band_blk", int(os.environ.get("BLK", "1"))); eng.set("force_kind", 2); eng.set("block", 512); eng.set("band_pack", 1)
    b = eng.upload(PackedBatch(pairs)); b.align(mw.opt_init()); s, it, _ = b.results()
    exp = [orc.align(t, q, make_opt())[:2] for t, q in pairs]
    print(tl, n, [(int(a), int(c)) for a, c in zip(s, it)], exp, "retries", eng.stats().n_retries, "block", eng.stats().block, flush=True)
    b.free(); eng.close()
