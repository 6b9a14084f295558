import sys, os
sys.path.insert(0, '.')
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
pk = PackedBatch([synth_pair(60000 + i, 50000, 0.03) for i in range(1250)])
ref = None
for ring16, blk in ((0, 512), (1, 768), (1, 512)):
    for flag in (0, 1):
        eng = mw.Engine(0); eng.set("ring16", ring16); eng.set("ring16_block", blk)
        b = eng.upload(pk); o = mw.opt_init(flag=flag)
        for _ in range(2): b.align(o); s, it, nc = b.results()
        st = eng.stats()
        key = (np.array(s).tobytes(), np.array(it).tobytes())
        if flag == 0:
            if ref is None: ref = key
        print(f"ring16={ring16} block {st.block} grid {st.grid} flag={flag}: kernel {st.kernel_ms:.1f} ms retries {st.n_retries} same_as_32bit {key == ref}", flush=True)
        b.free(); eng.close()
