import sys, time
sys.path.insert(0, '.')
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
import time
for n, tl in ((2, 50000), (4, 50000), (8, 50000), (16, 50000), (32, 50000), (64, 50000), (4, 100000), (8, 100000), (16, 100000), (3, 150000), (8, 150000)):
    pairs = [synth_pair(7000 + i, tl, 0.03) for i in range(n)]
    res = {}
    for kind in (-1, 1, 0):
        eng = mw.Engine(0)
        eng.set("force_kind", kind)
        b = eng.upload(PackedBatch(pairs))
        for flag in (0,):
            b.align(mw.opt_init(flag=flag)); s, it, nc = b.results()
            t0 = time.perf_counter(); b.align(mw.opt_init(flag=flag)); s, it, nc = b.results(); wall = (time.perf_counter() - t0) * 1e3
            res[kind] = (wall, eng.stats().kernel_kind, tuple(s.tolist()))
        b.free(); eng.close()
    print(n, "x", tl, "wall ms (kernel kind) by force_kind:", {k: (round(v[0], 1), v[1]) for k, v in res.items()}, "same:", len({v[2] for v in res.values()}) == 1, flush=True)
