"""Penalty sets other than the default under the default routing — the reference's presets (main.c:34-35: -e edit distance, -a one gap piece), minimap2-like and
large gap-open costs — on the headline batch, a read batch, a mid-size batch and one long pair: ms per align, which kernel, pairs run twice.
Usage (GPU box): python profiles/penalty_survey.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from miniwfa_amd import api as mw
from miniwfa_amd.synth import PackedBatch, synth_pair

PEN = [("default 4,4,2,24,1", {}), ("-a one piece 4,4,2", dict(o2=4, e2=2)), ("-e edit distance", dict(x=1, o1=0, e1=1, o2=0, e2=1)), ("6,6,2,30,1", dict(x=6, o1=6, e1=2, o2=30, e2=1)),
       ("2,4,2,24,1 (o1 != x)", dict(x=2, o1=4, e1=2, o2=24, e2=1)), ("4,6,2,60,1", dict(x=4, o1=6, e1=2, o2=60, e2=1)), ("5,8,2,100,2", dict(x=5, o1=8, e1=2, o2=100, e2=2)), ("4,4,3,24,1 (e1 = 3)", dict(x=4, o1=4, e1=3, o2=24, e2=1)), ("asm5-like 4,6,3,26,1", dict(x=4, o1=6, e1=3, o2=26, e2=1))]
if len(sys.argv) > 1:
    PEN = [x for x in PEN if "e1 = 3" in x[0] or "asm5" in x[0]]
SHAPES = [("1024 x 10 kb @ 5 %", [synth_pair(50000 + i, 10000, 0.05) for i in range(1024)]), ("20000 x 150 bp @ 5 %", [synth_pair(7000 + i, 150, 0.05) for i in range(20000)]),
          ("512 x 2 kb @ 5 %", [synth_pair(100 + i, 2000, 0.05) for i in range(512)]), ("100 x 1 kb @ 5 %", [synth_pair(100 + i, 1000, 0.05) for i in range(100)]), ("1 x 2 kb", [synth_pair(4242, 2000, 0.05)]), ("1 x 150 kb @ 3 %", [synth_pair(4242, 150000, 0.03)])]
for sname, pairs in SHAPES:
    pk = PackedBatch(pairs)
    for pname, kw in PEN:
        for flag in (0, 1):
            eng = mw.Engine(0)
            b = eng.upload(pk)
            o = mw.opt_init(flag=flag, **kw)
            ms = []
            for it in range(3):
                t0 = time.perf_counter(); b.align(o); b.results(); ms.append((time.perf_counter() - t0) * 1e3)
            st = eng.stats()
            print(f"{sname:22s} {pname:26s} {'cigar' if flag else 'score'}: {min(ms[1:]):9.3f} ms (kind {st.kernel_kind} block {st.block} packed {st.packed}, {st.n_retries} re-run, {st.n_launches} launches)", flush=True)
            b.free(); eng.close()
