"""Where a chain-mode call's time goes (MWF_CHAIN_TIMING=1: the library prints anchors / gaps / gap-fill batch / stitch per call) beside the compiled
reference's time for the same call.  Usage: MWF_CHAIN_TIMING=1 python profiles/chain_timing.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair
from oracle.pyoracle import Reference, make_opt
ref = Reference() if Reference.available() else None
for tl, p in ((30000, 0.04), (30000, 0.1), (5000, 0.05), (100000, 0.03)):
    t, q = synth_pair(4242, tl, p, 2, 800)
    for flag in (1, 0):
        for _ in range(3):
            t0 = time.perf_counter(); s, _, cig = mw.wfa_chain(t, q, mw.opt_init(flag=flag)); d = time.perf_counter() - t0
        line = f"{tl} bp @ {p} flag {flag}: this library {1e3 * d:.2f} ms (s = {s})"
        if ref:
            t0 = time.perf_counter(); es, _, ecig = ref.chain(t, q, make_opt(flag=flag)); dr = time.perf_counter() - t0
            line += f", reference {1e3 * dr:.2f} ms, equal {s == es and (None if cig is None else list(cig)) == ecig}"
        print(line, flush=True)
