import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch, skewed_pairs
from oracle.pyoracle import Oracle, make_opt
orc = Oracle()
pairs = [p for i, p in enumerate(skewed_pairs(1, 400, 200, 3000)) if max(len(p[0]), len(p[1])) >= 1500]
for idx in (0, 3, 5, 45):
    t, q = pairs[idx]
    for blk in (1, 0):
        eng = mw.Engine(0); eng.set("band_blk", blk); eng.set("force_kind", 2); eng.set("block", 512); eng.set("band_pack", 1)
        b = eng.upload(PackedBatch([(t, q)]))
        o = mw.opt_init()
        dev = b.debug_band(o, 0) - 1 - len(t)
        s, it, _ = b.results()
        ref = np.array(orc.band_trace(t, q, make_opt()), dtype=np.int32).reshape(-1, 2)
        n = min(len(dev), len(ref))
        bad = np.nonzero((dev[:n] != ref[:n]).any(axis=1))[0]
        es = orc.align(t, q, make_opt())
        print(f"pair {idx} ({len(t)} x {len(q)}) blk {blk}: got s {int(s[0])} n_iter {int(it[0])} expected {es[0]} {es[1]}; trace lens {len(dev)}/{len(ref)}; first divergence", (int(bad[0]) + 1, dev[bad[0]].tolist(), ref[bad[0]].tolist(), "prev", dev[bad[0]-1].tolist()) if len(bad) else None, flush=True)
        b.free(); eng.close()
