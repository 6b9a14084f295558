"""The lane kernel's LDS rows are allocated for `lane_chunks` 64-column chunks whatever the window does, and LDS bounds its waves per CU:
fewer chunks = more resident waves but more pairs handed back.  Kernel ms / step ms / re-runs per chunk count on read-length batches."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
for n, tl, div in ((40000, 150, 0.05), (20000, 250, 0.05), (40000, 150, 0.02), (40000, 100, 0.05)):
    pk = PackedBatch([synth_pair(7000 + i, tl, div) for i in range(n)])
    for ch in (0, 2, 1):
        eng = mw.Engine(0); eng.set("lane_chunks", ch)
        b = eng.upload(pk); o = mw.opt_init()
        km, wm = [], []
        for _ in range(6):
            t0 = time.perf_counter(); b.align(o); b.results(); wm.append((time.perf_counter() - t0) * 1e3); km.append(eng.stats().kernel_ms)
        st = eng.stats()
        print(f"{n} x {tl} @ {div:g}, lane_chunks {ch}: kernel {np.median(km[2:]):.3f} ms, step {np.median(wm[2:]):.3f} ms, re-run {st.n_retries}", flush=True)
        b.free(); eng.close()
