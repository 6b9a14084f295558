"""16-bit ring rows (packed recurrence, 2-bit sequence copies) against 32-bit rows on batches SMALLER than the number of CUs, where
round 2 measured the coding as pure overhead (64 x 50 kb: 88 ms against 73): n pairs of 50 kb @ 3 %, kernel ms of the second call."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
allp = [synth_pair(60000 + i, 50000, 0.03) for i in range(256)]
for n in (8, 32, 64, 128, 200, 256):
    pk = PackedBatch(allp[:n]); ref = None
    for ring16 in (0, 2):
        for flag in (0, 1):
            eng = mw.Engine(0); eng.set("ring16", ring16); eng.set("force_kind", 0)
            b = eng.upload(pk); o = mw.opt_init(flag=flag)
            for _ in range(2): b.align(o); s, it, nc = b.results()
            st = eng.stats(); key = (np.array(s).tobytes(), np.array(it).tobytes())
            if ref is None: ref = key
            print(f"{n} pairs ring16={ring16} flag={flag}: block {st.block} grid {st.grid} packed {st.packed} kernel {st.kernel_ms:.1f} ms retries {st.n_retries} same {key == ref}", flush=True)
            b.free(); eng.close()
