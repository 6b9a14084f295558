"""Batches of other size classes of the packed band kernel (biased five-slot 512-thread copies, 256- and 128-thread geometries), kernel time per align:
compare a library built with -DMWF_B2_MERGE_OFF=1 (profiles/build_band2_variant.sh) against the default.  Usage: [MWF_HIP_LIB=...] python profiles/merge_probe.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
for n, tl, p in ((1024, 15000, 0.04), (2048, 5000, 0.05), (8192, 2000, 0.05), (16384, 800, 0.05)):
    pk = PackedBatch([synth_pair(31000 + i, tl, p) for i in range(n)])
    for flag in (0, 1):
        eng = mw.Engine(0); b = eng.upload(pk); o = mw.opt_init(flag=flag)
        ks = []
        for _ in range(6):
            b.align(o); b.results(); ks.append(eng.stats().kernel_ms)
        st = eng.stats()
        print(f"{n} x {tl} @ {p} flag {flag}: kernel {np.median(ks[2:]):.3f} ms (block {st.block}, re-run {st.n_retries})", flush=True)
        b.free(); eng.close()
