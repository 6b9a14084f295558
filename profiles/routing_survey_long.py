"""Batches of long pairs under the default routing (between the span geometry's 62 kb and the whole-device kernel's few pairs): ms per align of a resident
batch, pairs run twice, which kernel.  Usage (GPU box): python profiles/routing_survey_long.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from miniwfa_amd import api as mw
from miniwfa_amd.synth import PackedBatch, synth_pair

S = [("32 x 60 kb @ 3 %", 32, 60000, 0.03), ("32 x 100 kb @ 3 %", 32, 100000, 0.03), ("100 x 100 kb @ 2 %", 100, 100000, 0.02), ("20 x 150 kb @ 3 %", 20, 150000, 0.03),
     ("4 x 500 kb @ 2 %", 4, 500000, 0.02), ("300 x 70 kb @ 3 %", 300, 70000, 0.03), ("16 x 30 kb @ 5 %", 16, 30000, 0.05), ("17 x 30 kb @ 5 %", 17, 30000, 0.05), ("40 x 30 kb @ 5 %", 40, 30000, 0.05), ("64 x 30 kb @ 5 %", 64, 30000, 0.05),
     ("36 x 100 kb @ 3 %", 36, 100000, 0.03), ("48 x 100 kb @ 3 %", 48, 100000, 0.03), ("24 x 20 kb @ 5 %", 24, 20000, 0.05)]
if len(sys.argv) > 1:
    S = [x for x in S if x[0] in sys.argv[1:]]
for name, n, L, p in S:
    pairs = [synth_pair(900 + i, L, p) for i in range(n)]
    pk = PackedBatch(pairs)
    bp = sum(len(t) + len(q) for t, q in pairs)
    for mode, kw in (("score", dict(flag=0)), ("cigar", dict(flag=1)), ("low-mem", dict(flag=1, step=5000))):
        eng = mw.Engine(0)
        b = eng.upload(pk)
        o = mw.opt_init(**kw)
        rec = []
        for it in range(2):
            t0 = time.perf_counter(); b.align(o); b.results(); rec.append((time.perf_counter() - t0) * 1e3)
        st = eng.stats()
        print(f"{name:22s} {mode:8s}: first {rec[0]:9.2f} ms, then {rec[1]:9.2f} ms (kernels {st.kernel_ms:8.2f}, {st.n_retries} re-run, {st.n_launches} launches, kind {st.kernel_kind} block {st.block} packed {st.packed} two-pass {st.lowmem_two_pass}), {bp / rec[1] / 1e6:6.3f} Gbp/s, {st.dev_bytes_peak / 2**30:6.1f} GB", flush=True)
        b.free(); eng.close()
