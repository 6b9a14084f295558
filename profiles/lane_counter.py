"""What a launch of the lane kernel costs besides the alignments: 40 000 x 150 bp pairs at divergences 0 (every pair ends at penalty 0) to 8 %,
and batches of other sizes at 5 %.  profiles/r04/lane_counter.txt holds the table with ONE global work counter (through round 4: ~12.7 ns per
pair whatever the reads), with a static deal of the pairs, and with the 64 partitioned counters the kernel uses now."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
for n, tl, div in ((40000, 150, 0.0), (40000, 150, 0.01), (40000, 150, 0.02), (40000, 150, 0.05), (40000, 150, 0.08), (10000, 150, 0.05), (2816, 150, 0.05), (80000, 150, 0.05)):
    pk = PackedBatch([synth_pair(7000 + i, tl, div) for i in range(n)])
    eng = mw.Engine(0)
    b = eng.upload(pk); o = mw.opt_init()
    km = []
    for _ in range(6):
        b.align(o); s, it, _ = b.results(); km.append(eng.stats().kernel_ms)
    st = eng.stats()
    print(f"{n} x {tl} @ {div:g}: kernel {np.median(km[2:]):.3f} ms, mean s {s.mean():.1f}, cells {int(it.sum())}, grid {st.grid} re-run {st.n_retries}", flush=True)
    b.free(); eng.close()
