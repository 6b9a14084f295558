"""CIGAR batch (1024 x 10 kb @ 5 %): unpacked 768x2 band kernel against the int16-packed 768x2 one."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
pairs = [synth_pair(50000 + i, 10000, 0.05) for i in range(1024)]
pk = PackedBatch(pairs)
ref = None
for pack in (0, 1, -1):
    eng = mw.Engine(0)
    eng.set("band_pack", pack)
    b = eng.upload(pk)
    o = mw.opt_init(flag=1)
    ms = []
    for _ in range(4):
        b.align(o); s, it, nc = b.results(); ms.append(eng.stats().kernel_ms)
    cg = [b.cigar(i, int(nc[i])).tobytes() for i in range(0, 1024, 37)]
    cur = (s.tobytes(), it.tobytes(), nc.tobytes(), cg)
    if ref is None: ref = cur
    print(f"band_pack={pack}: kernel_ms={min(ms):.2f} block={eng.stats().block} grid={eng.stats().grid} same={cur == ref}")
    b.free(); eng.close()
