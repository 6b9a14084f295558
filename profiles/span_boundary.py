"""Where the 512-thread geometry (two pairs per CU, 24 chunks) hands over to the 1024-thread span geometry (one per CU, 80 chunks):
score-only batches of mid-size pairs on default settings, with "band_span" 2 (every pair on the span geometry) and 0 (never)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
for n, tl, div in ((1024, 10000, 0.05), (1024, 12000, 0.05), (512, 15000, 0.05), (512, 16000, 0.03), (512, 24000, 0.03), (256, 30000, 0.03), (512, 12000, 0.02)):
    pk = PackedBatch([synth_pair(60000 + i, tl, div) for i in range(n)])
    out = []
    for span in (1, 2, 0):
        eng = mw.Engine(0); eng.set("band_span", span)
        b = eng.upload(pk); o = mw.opt_init()
        b.align(o); b.results()
        t0 = time.perf_counter(); b.align(o); s, it, _ = b.results(); wall = (time.perf_counter() - t0) * 1e3
        st = eng.stats()
        out.append(f"span={span}: {wall:.1f} ms (first launch {st.kernel_ms:.1f}, re-run {st.n_retries}, last {st.kernel_kind}/{st.packed}/{st.block})")
        b.free(); eng.close()
    print(f"{n} x {tl} @ {div:g} s~{int(s.mean())}: " + " | ".join(out), flush=True)
