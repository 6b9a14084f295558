"""One-shot, host-to-host cost of a batch on a WARM engine (pinned staging and device blocks pooled): upload (pack + H2D), first align, results,
free — what a user who aligns a batch once pays.  Usage: [MWF_UPLOAD_TIMING=1] python profiles/one_shot_probe.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
eng = mw.Engine(0)
for n, tl, p in ((40000, 150, 0.05), (20000, 250, 0.05), (1024, 10000, 0.05)):
    pks = [PackedBatch([synth_pair(7000 + 100000 * r + i, tl, p) for i in range(n)]) for r in range(5)]
    o = mw.opt_init()
    rows = []
    for r, pk in enumerate(pks):
        t0 = time.perf_counter(); b = eng.upload(pk); t1 = time.perf_counter(); b.align(o); t2 = time.perf_counter(); s, it, _ = b.results(); t3 = time.perf_counter(); b.free(); t4 = time.perf_counter()
        rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t0) * 1e3, eng.stats().n_retries, pk.bases))
    for r, x in enumerate(rows):
        print(f"{n} x {tl} bp @ {p}: batch {r}: upload {x[0]:.2f} ms, align (enqueue) {x[1]:.2f}, results {x[2]:.2f}, free {x[3]:.2f}, total {x[4]:.2f} ms = {x[6] / x[4] / 1e6:.2f} Gbp/s host to host, re-run {x[5]}", flush=True)
eng.close()
