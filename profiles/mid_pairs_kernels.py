"""Pairs too long for the packed band kernel (target + penalty bound >= 32767) but far from the whole-device kernel's sizes: the default
choice against the generic kernel with 16-bit ring rows and against the unpacked band kernel, wall ms of the second call (align + results, retries included), results compared."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
for tl, div, n in ((12000, 0.05, 1024), (20000, 0.03, 512), (20000, 0.01, 512), (30000, 0.005, 512), (50000, 0.003, 512)):
    pk = PackedBatch([synth_pair(70000 + i, tl, div) for i in range(n)]); ref = None
    for name, sets in (("default", {}), ("generic (16-bit rows)", {"force_kind": 0}), ("band, unpacked", {"force_kind": 2})):
        for flag in (0, 1):
            eng = mw.Engine(0)
            for k, v in sets.items(): eng.set(k, v)
            b = eng.upload(pk); o = mw.opt_init(flag=flag)
            for _ in range(2):
                t0 = time.perf_counter(); b.align(o); s, it, nc = b.results(); wall = time.perf_counter() - t0
            st = eng.stats(); key = (np.array(s).tobytes(), np.array(it).tobytes())
            if ref is None: ref = key
            print(f"{n} x {tl} @ {div} {name} flag={flag}: kind {st.kernel_kind} block {st.block} packed {st.packed} grid {st.grid} wall {wall * 1e3:.1f} ms (last launch {st.kernel_ms:.1f}) retries {st.n_retries} same {key == ref}", flush=True)
            b.free(); eng.close()
