"""Lane kernel (one wave per pair, mwf_lane.hip) against the 64-thread geometry of the packed band kernel on read batches by length: where
`lane_max_len` should sit.  ms per align + results of a resident batch (third align), pairs run twice.
Usage (GPU box): python profiles/lane_crossover.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from miniwfa_amd import api as mw
from miniwfa_amd.synth import PackedBatch, synth_pair

for n in (2000, 20000):
    for L in (150, 200, 250, 300, 350, 400, 450):
        for p in (0.05,) if n == 20000 else (0.02, 0.05, 0.10):
            pairs = [synth_pair(33000 + i, L, p) for i in range(n)]
            pk = PackedBatch(pairs)
            out = []
            for flag in (0, 1):
                for lane in (1000, 0):
                    eng = mw.Engine(0)
                    eng.set("lane_max_len", lane)
                    b = eng.upload(pk)
                    o = mw.opt_init(flag=flag)
                    for it in range(3):
                        t0 = time.perf_counter(); b.align(o); b.results(); ms = (time.perf_counter() - t0) * 1e3
                    st = eng.stats()
                    out.append(f"{'cigar' if flag else 'score'} {'lane' if lane else 'band'} {ms:6.3f} ms ({st.n_retries} re-run, packed {st.packed})")
                    b.free(); eng.close()
            print(f"{n} x {L} bp @ {p}: " + " | ".join(out), flush=True)
