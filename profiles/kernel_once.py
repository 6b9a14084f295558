"""One workload on one kernel, a few aligns (for rocprofv3 / PMC passes: profiles/pmc_cmd.sh <tag> python profiles/kernel_once.py <what>).
what: lane (40 000 x 150 bp @ 5 %, score) | mid1 (one 2 kb pair @ 5 %, score) | mid256 (256 x 2 kb @ 5 %, score)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
what = sys.argv[1]
n, tl = {"lane": (40000, 150), "mid1": (1, 2000), "mid256": (256, 2000)}[what]
pairs = [synth_pair(7000 + i, tl, 0.05) for i in range(n)]
pk = PackedBatch(pairs)
eng = mw.Engine(0)
b = eng.upload(pk)
for rep in range(4):
    b.align(mw.opt_init()); s, it, nc = b.results()
st = eng.stats()
print(f"{what}: {n} x {tl} bp, cells/launch {int(it.sum())}, kernel {st.kernel_ms:.4f} ms, packed {st.packed} block {st.block} grid {st.grid} re-run {st.n_retries}", flush=True)
b.free(); eng.close()
