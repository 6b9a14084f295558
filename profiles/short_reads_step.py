"""Read-sized batches end to end: kernel time (HIP events) and wall time of align() + results() with the batch resident, first align
(the per-pair classification runs) and repeated aligns (the plan of the last align applies); score-only and CIGAR.
Usage: python profiles/short_reads_step.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch

for n, tl, p in ((40000, 150, 0.05), (20000, 250, 0.05), (20000, 150, 0.10), (20000, 300, 0.02)):
    pairs = [synth_pair(7000 + i, tl, p) for i in range(n)]
    bp = sum(len(t) + len(q) for t, q in pairs)
    pk = PackedBatch(pairs)
    for label, kw in (("score", {}), ("cigar", {"flag": 1})):
        for s2 in (1, 0):
            eng = mw.Engine(0)
            eng.set("seq2bit", s2)
            t0 = time.perf_counter(); b = eng.upload(pk); up = (time.perf_counter() - t0) * 1e3
            o = mw.opt_init(**kw)
            t0 = time.perf_counter(); b.align(o); b.results(); first = (time.perf_counter() - t0) * 1e3
            ks, ws = [], []
            for _ in range(7):
                t0 = time.perf_counter(); b.align(o); b.results(); ws.append((time.perf_counter() - t0) * 1e3)
                ks.append(eng.stats().kernel_ms)
            st = eng.stats()
            print(f"{n} x {tl} bp @ {p:.2f} {label} {'2bit ' if s2 else 'bytes'}: upload {up:.2f} ms, first align+results {first:.2f} ms, then step {np.median(ws):.3f} ms "
                  f"({bp / np.median(ws) / 1e6:.2f} Gbp/s), kernel {np.median(ks):.3f} ms ({bp / np.median(ks) / 1e6:.2f} Gbp/s), re-run {st.n_retries}", flush=True)
            b.free(); eng.close()
