import sys, time
sys.path.insert(0, '.')
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair
for tl in (200, 2000, 10000):
    t, q = synth_pair(123, tl, 0.05)
    o = mw.opt_init()
    for _ in range(3): mw.wfa_exact(t, q, o)
    n = 50; t0 = time.perf_counter()
    for _ in range(n): mw.wfa_exact(t, q, o)
    print(tl, "score-only per call ms", (time.perf_counter() - t0) / n * 1e3)
    o = mw.opt_init(flag=1)
    for _ in range(3): mw.wfa_exact(t, q, o)
    t0 = time.perf_counter()
    for _ in range(n): mw.wfa_exact(t, q, o)
    print(tl, "cigar per call ms", (time.perf_counter() - t0) / n * 1e3)
