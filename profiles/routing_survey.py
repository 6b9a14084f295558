"""Batch shapes under the default routing: ms per align of a resident batch (second align: cached plan), pairs run twice, launches, Gbp/s —
a look for shapes the host's classes serve badly (pairs run twice, a launch that lasts as long as one straggler).
Usage (GPU box): python profiles/routing_survey.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from miniwfa_amd import api as mw
from miniwfa_amd.synth import PackedBatch, synth_pair


def shape(name, pairs, flag=0):
    eng = mw.Engine(0)
    pk = PackedBatch(pairs)
    b = eng.upload(pk)
    o = mw.opt_init(flag=flag)
    rec = []
    for it in range(3):
        t0 = time.perf_counter()
        b.align(o)
        b.results()
        ms = (time.perf_counter() - t0) * 1e3
        st = eng.stats()
        rec.append((ms, st.kernel_ms, st.n_retries, st.n_launches))
    bp = sum(len(t) + len(q) for t, q in pairs)
    ms, kms, rr, nl = rec[-1]
    print(f"{name:44s} flag {flag}: first {rec[0][0]:8.3f} ms ({rec[0][2]} re-run), then {ms:8.3f} ms (kernels {kms:7.3f}, {rr} re-run, {nl} launches), {bp / ms / 1e6:6.3f} Gbp/s", flush=True)
    b.free()
    eng.close()


rng = np.random.default_rng(5)
S = [
    ("512 x 3 kb @ 5 %", [synth_pair(100 + i, 3000, 0.05) for i in range(512)]),
    ("2000 x 500 bp @ 5 %", [synth_pair(2000 + i, 500, 0.05) for i in range(2000)]),
    ("300 x 1 kb @ 5 %", [synth_pair(5000 + i, 1000, 0.05) for i in range(300)]),
    ("200 x 1 kb @ 5 %", [synth_pair(5000 + i, 1000, 0.05) for i in range(200)]),
    ("4000 x 1 kb @ 10 %", [synth_pair(6000 + i, 1000, 0.10) for i in range(4000)]),
    ("128 x 5 kb @ 5 %", [synth_pair(11000 + i, 5000, 0.05) for i in range(128)]),
    ("64 x 20 kb @ 3 %", [synth_pair(12000 + i, 20000, 0.03) for i in range(64)]),
    ("600 x 20 kb @ 3 %", [synth_pair(12000 + i, 20000, 0.03) for i in range(600)]),
    ("1000 log-uniform 100 .. 20000 @ 5 %", [synth_pair(13000 + i, int(np.exp(rng.uniform(np.log(100), np.log(20000)))), 0.05) for i in range(1000)]),
    ("3000 reads 150 bp + 40 x 2 kb", [synth_pair(17000 + i, 150, 0.05) for i in range(3000)] + [synth_pair(21000 + i, 2000, 0.05) for i in range(40)]),
    ("1024 x 10 kb + 100 x 2 kb", [synth_pair(50000 + i, 10000, 0.05) for i in range(1024)] + [synth_pair(21000 + i, 2000, 0.05) for i in range(100)]),
    ("2000 x 300 bp @ 5 %", [synth_pair(31000 + i, 300, 0.05) for i in range(2000)]),
    ("2000 x 450 bp @ 5 %", [synth_pair(33000 + i, 450, 0.05) for i in range(2000)]),
    ("1000 x 1.5 kb @ 2 %", [synth_pair(35000 + i, 1500, 0.02) for i in range(1000)]),
]
for name, pairs in S:
    for flag in (0, 1):
        shape(name, pairs, flag)
