"""1024 x 10 kb @ 5 % batches of four seeds, six aligns each (median of the last four): a pair that outgrows the 512-thread geometry late is re-run alone."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
for seed in (50000, 60000, 70000, 80000):
    pk = PackedBatch([synth_pair(seed + i, 10000, 0.05) for i in range(1024)])
    eng = mw.Engine(0)
    if len(sys.argv) > 1: eng.set("band_fold", int(sys.argv[1]))  # 0: rows of all three lags from HBM, 1: folded form, 2: and the last penalties' rows in LDS (default)
    if len(sys.argv) > 3 and sys.argv[3] == "span": eng.set("band_span", 2)  # third argument "span": every pair on the 1024-thread span geometry
    b = eng.upload(pk); o = mw.opt_init(flag=int(sys.argv[2]) if len(sys.argv) > 2 else 0)  # second argument 1: with CIGAR
    w = []
    for _ in range(6):
        t0 = time.perf_counter(); b.align(o); s, it, _ = b.results(); w.append((time.perf_counter() - t0) * 1e3)
    st = eng.stats()
    print(f"seed {seed}: step {np.median(w[2:]):.2f} ms, re-run {st.n_retries}, s max {int(s.max())}, checksum {int(s.astype(np.int64).sum())} {int(it.astype(np.int64).sum())}", flush=True)
    b.free(); eng.close()
