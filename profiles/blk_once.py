"""The headline batch (1024 x 10 kb @ 5 %, one seed) on the per-penalty (0) or block (1) form of the packed band kernel, four aligns — for PMC passes:
profiles/pmc_cmd.sh <tag> python profiles/blk_once.py <0|1> [pairs]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
blk = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
pk = PackedBatch([synth_pair(60000 + i, 10000, 0.05) for i in range(n)])
eng = mw.Engine(0); eng.set("band_blk", blk); eng.set("wide_slots", 4)
b = eng.upload(pk)
for rep in range(4):
    b.align(mw.opt_init()); s, it, nc = b.results()
st = eng.stats()
print(f"blk {blk}: {n} x 10 kb, cells/launch {int(it.sum())}, kernel {st.kernel_ms:.4f} ms, block {st.block} grid {st.grid} re-run {st.n_retries}", flush=True)
b.free(); eng.close()
