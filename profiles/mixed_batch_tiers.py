"""A batch of mixed lengths through the drop-in mwf_wfa_batch as one call, against the same pairs cut into length tiers that are aligned by concurrent calls
(one host thread and one pooled engine each, all on the same device): do the size classes of one call, which run one after the other on one stream, leave
the device idle?  Usage (GPU box): python profiles/mixed_batch_tiers.py"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from miniwfa_amd import api as mw
from miniwfa_amd.synth import synth_pair

rng = np.random.default_rng(5)
pairs = [synth_pair(13000 + i, int(np.exp(rng.uniform(np.log(100), np.log(20000)))), 0.05) for i in range(1000)]
bp = sum(len(t) + len(q) for t, q in pairs)


def one_call(flag):
    return mw.wfa_batch(pairs, mw.opt_init(flag=flag))


def tiers(flag, cuts):
    idx = sorted(range(len(pairs)), key=lambda i: len(pairs[i][0]) + len(pairs[i][1]))
    groups, lo = [], 0
    for c in list(cuts) + [1 << 40]:
        g = [i for i in idx[lo:] if len(pairs[i][0]) + len(pairs[i][1]) <= c]
        lo += len(g)
        if g:
            groups.append(g)
    out = [None] * len(pairs)

    def work(g):
        r = mw.wfa_batch([pairs[i] for i in g], mw.opt_init(flag=flag))
        for i, x in zip(g, r):
            out[i] = x
    th = [threading.Thread(target=work, args=(g,)) for g in groups]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return out


for flag in (0, 1):
    ref = None
    for name, fn in (("one call", lambda: one_call(flag)), ("tiers 3000 / 12000", lambda: tiers(flag, (3000, 12000))), ("tiers 2000 / 8000 / 20000", lambda: tiers(flag, (2000, 8000, 20000))),
                     ("tiers 8000", lambda: tiers(flag, (8000,))), ("tiers 1000 / 4000 / 10000 / 24000", lambda: tiers(flag, (1000, 4000, 10000, 24000)))):
        ms = []
        for it in range(4):
            t0 = time.perf_counter()
            r = fn()
            ms.append((time.perf_counter() - t0) * 1e3)
        key = [(x[0], x[1]) for x in r]
        if ref is None:
            ref = key
        print(f"flag {flag} {name:36s}: {min(ms[1:]):8.3f} ms ({bp / min(ms[1:]) / 1e6:.3f} Gbp/s), same answers {key == ref}", flush=True)
