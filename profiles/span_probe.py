"""The 1024-thread span geometry of the packed band kernel against the generic kernel (16-bit ring rows) on long pairs.
  python profiles/span_probe.py parity            -- a few 20-50 kb pairs, score / n_iter / CIGAR against the compiled oracle
  python profiles/span_probe.py time [n] [tl] [div]  -- n x tl @ div, score-only: band_span 1 against 0 (kernel ms incl. re-runs = wall)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch

what = sys.argv[1]
if what == "parity":
    import pyoracle
    O = pyoracle.Oracle()
    specs = [(50000, 0.03), (50000, 0.02), (40000, 0.035), (33000, 0.04), (20000, 0.06), (60000, 0.015), (50000, 0.034), (25000, 0.05)]
    pairs = [synth_pair(910 + i, tl, d) for i, (tl, d) in enumerate(specs)]
    pk = PackedBatch(pairs)
    eng = mw.Engine(0)
    b = eng.upload(pk)
    for flag in (0, mw.MWF_F_CIGAR):
        o = mw.opt_init(); o.flag = flag
        b.align(o); s, it, nc = b.results()
        st = eng.stats()
        print(f"flag {flag}: kind {st.kernel_kind} packed {st.packed} block {st.block} re-run {st.n_retries} kernel {st.kernel_ms:.2f} ms", flush=True)
        oo = O.opt_init(); oo.flag = flag
        for i, (t, q) in enumerate(pairs):
            rs, rit, rcig = O.align(t, q, oo)
            ok = rs == s[i] and rit == it[i]
            if flag:
                cg = b.cigar(i, int(nc[i]))
                ok = ok and list(cg) == list(rcig)
            print(f"  pair {i} tl {len(t)} s {s[i]} (oracle {rs}) n_iter {it[i]} ({rit}) {'ok' if ok else 'MISMATCH'}", flush=True)
    b.free(); eng.close()
else:
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1250
    tl = int(sys.argv[3]) if len(sys.argv) > 3 else 50000
    div = float(sys.argv[4]) if len(sys.argv) > 4 else 0.03
    pairs = [synth_pair(60000 + i, tl, div) for i in range(n)]
    pk = PackedBatch(pairs)
    res = {}
    for span in (1, 0):
        eng = mw.Engine(0)
        eng.set("band_span", span)
        b = eng.upload(pk)
        o = mw.opt_init()
        b.align(o); b.results()
        t0 = time.perf_counter()
        b.align(o); s, it, _ = b.results()
        wall = (time.perf_counter() - t0) * 1e3
        st = eng.stats()
        res[span] = (s.copy(), it.copy())
        print(f"{n} x {tl} @ {div:g} band_span {span}: wall {wall:.1f} ms, first-launch kernel {st.kernel_ms:.1f} ms, last kind {st.kernel_kind} packed {st.packed} block {st.block}, re-run {st.n_retries}, "
              f"cells {int(it.sum())}, s max {int(s.max())} mean {s.mean():.0f}", flush=True)
        b.free(); eng.close()
    print("equal:", bool((res[0][0] == res[1][0]).all() and (res[0][1] == res[1][1]).all()))
