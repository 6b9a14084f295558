"""Where the kernel chooser's thresholds fray: re-run rates per (length, divergence, batch size) on DEFAULT settings.  A pair that
outgrows the span of the kernel it was given (lane: 64 x chunks columns; mid: its LDS span; band classes: their register span)
comes back and is re-run on the next wider kernel — wasted work.  Prints kernel (stats.packed: 32 lane, 33 mid, 1 packed band,
16 generic with 16-bit rows, 0 other; block — 1024 with packed 1 is the packed band kernel's span geometry), re-runs / pairs, step time; marks classes above 2 %.
Usage: python profiles/chooser_regression.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch

print("# length  div   pairs | kernel(packed,block,kind) | re-runs (rate) | step ms")
worst = []
for tl in (150, 300, 400, 1000, 2000, 3000, 6000, 10000, 20000, 50000):
    for p in (0.01, 0.05, 0.15, 0.30):
        for n in (1, 64, 2000 if tl <= 3000 else 600 if tl <= 10000 else 256):
            pairs = [synth_pair(330000 + 977 * i + tl, tl, p) for i in range(n)]
            pk = PackedBatch(pairs)
            eng = mw.Engine(0)
            b = eng.upload(pk)
            o = mw.opt_init()
            b.align(o); b.results()
            t0 = time.perf_counter(); b.align(o); s, it, _ = b.results(); dt = (time.perf_counter() - t0) * 1e3
            st = eng.stats()
            rate = st.n_retries / n
            mark = "  <-- above 2 %" if rate > 0.02 else ""
            if rate > 0.02:
                worst.append((tl, p, n, rate))
            print(f"{tl:6d} {p:5.2f} {n:6d} | ({st.packed},{st.block},{st.kernel_kind}) | {st.n_retries:5d} ({100 * rate:5.1f} %) | {dt:8.3f}{mark}", flush=True)
            b.free(); eng.close()
print("# classes above 2 % re-runs:", worst)
