"""Generic kernel, two-pass low-memory mode on a batch of medium pairs: four columns per lane (both passes) against the
one-column-per-lane passes (scalar_generic = 1)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
pairs = [synth_pair(97000 + i, 20000, 0.04) for i in range(256)]
pk = PackedBatch(pairs)
ref = None
for scalar in (0, 1):
    eng = mw.Engine(0); eng.set("force_kind", 0); eng.set("scalar_generic", scalar)
    b = eng.upload(pk)
    ms = []
    for _ in range(3):
        b.align(mw.opt_init(flag=1, step=1000)); s, it, nc = b.results(); ms.append(eng.stats().kernel_ms)
    st = eng.stats()
    print(f"scalar_generic={scalar}: kernel ms {[round(x, 1) for x in ms]} block {st.block} grid {st.grid} cells pass1 {st.cells_pass1:.3e} pass2 {st.cells:.3e}", flush=True)
    cur = (s.tolist(), it.tolist())
    assert ref is None or ref == cur
    ref = cur
    b.free(); eng.close()
