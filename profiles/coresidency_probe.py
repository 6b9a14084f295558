"""Headline batch (1024 x 10 kb @ 5 %, score-only) on the packed band kernel with a chosen number of workgroups per CU
(argv[1]: 0 = the kernel's occupancy (two), 1 = one per CU), a few aligns: for rocprofv3 --pmc passes
(profiles/r05_coresidency.sh) that attribute the co-residency penalty.  argv[2] (optional): number of pairs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
per_cu = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
pk = PackedBatch([synth_pair(50000 + i, 10000, 0.05) for i in range(n)])
eng = mw.Engine(0)
if per_cu:
    eng.set("slots_per_cu", per_cu)
eng.set("wide_slots", 3)
b = eng.upload(pk); o = mw.opt_init()
w = []
for _ in range(4):
    t0 = time.perf_counter(); b.align(o); s, it, _ = b.results(); w.append((time.perf_counter() - t0) * 1e3)
st = eng.stats()
print(f"per_cu {per_cu} pairs {n}: step {np.median(w[1:]):.3f} ms kernel {st.kernel_ms:.3f} ms grid {st.grid} block {st.block} cells {int(it.sum())} re-run {st.n_retries}", flush=True)
b.free(); eng.close()
