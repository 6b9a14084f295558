import sys; sys.path.insert(0, '.')
import torch, numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
def run(pairs, kw, lane):
    eng = mw.Engine(0); eng.set("lane_max_len", lane)
    b = eng.upload(PackedBatch(pairs)); o = mw.opt_init(**kw)
    for _ in range(2): b.align(o); b.results()
    ks = []
    import time; t0 = time.perf_counter()
    for _ in range(5): b.align(o); r = b.results(); ks.append(eng.stats().kernel_ms)
    wall = (time.perf_counter() - t0) / 5 * 1e3
    st = eng.stats(); b.free(); eng.close()
    return wall, sum(ks) / 5, st.n_retries, r[0]
for n, ln, div in ((20000, 350, 0.05), (20000, 400, 0.05), (20000, 400, 0.03), (20000, 500, 0.02)):
    pairs = [synth_pair(9000 + i, ln, div) for i in range(n)]
    for kw in (dict(), dict(flag=1)):
        w0, k0, r0, s0 = run(pairs, kw, 0)
        w1, k1, r1, s1 = run(pairs, kw, 520)
        assert (s0 == s1).all()
        print(f"{n} x {ln} @ {div}: {'cigar' if kw else 'score'} band {w0:.2f} wall {k0:.2f} kernel | lane {w1:.2f} wall {k1:.2f} first launch {r1} re-run", flush=True)
