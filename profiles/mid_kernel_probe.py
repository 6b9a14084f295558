"""The mid kernel (mwf_mid.hip) against the packed band kernel on single pairs and small batches: per-call wall time of mwf_wfa_exact
and HIP-event kernel time, score-only / CIGAR.  Usage: python profiles/mid_kernel_probe.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch

def timed(eng, pk, o, reps=8):
    b = eng.upload(pk)
    for _ in range(2):
        b.align(o); b.results()
    ms, wall = [], []
    for _ in range(reps):
        t0 = time.perf_counter(); b.align(o); r = b.results(); wall.append((time.perf_counter() - t0) * 1e3)
        ms.append(eng.stats().kernel_ms)
    st = eng.stats()
    b.free()
    return float(np.median(ms)), float(np.median(wall)), st.packed, st.block, st.n_retries, r

print("# pairs x length @ divergence | mode | mid kernel: kernel ms, step ms (packed code, block, re-runs) | band kernels: kernel ms, step ms (code, block)")
for n, tl, p in ((1, 500, 0.05), (1, 1000, 0.05), (1, 2000, 0.05), (1, 2000, 0.1), (1, 3000, 0.05), (1, 4000, 0.05), (1, 5000, 0.03),
                 (64, 2000, 0.05), (256, 2000, 0.05), (256, 1000, 0.05), (512, 2000, 0.05)):
    pairs = [synth_pair(4000 + i, tl, p) for i in range(n)]
    pk = PackedBatch(pairs)
    for label, kw in (("score", {}), ("cigar", {"flag": 1})):
        o = mw.opt_init(**kw)
        outs = []
        res = []
        for mid_pairs, blk in ((1 << 20, 1024), (1 << 20, 512), (1 << 20, 256), (0, 0)):
            eng = mw.Engine(0)
            eng.set("mid_max_pairs", mid_pairs)
            eng.set("mid_block", blk)
            k, w, code, block, rr, r = timed(eng, pk, o)
            outs.append(f"{k:8.3f} {w:8.3f} ({code},{block},{rr})")
            res.append(r)
            eng.close()
        same = all((res[0][0] == r[0]).all() and (res[0][1] == r[1]).all() for r in res[1:])
        print(f"{n:5d} x {tl:5d} @ {p:.2f} {label:5s} | mid1024 {outs[0]} | mid512 {outs[1]} | mid256 {outs[2]} | band {outs[3]} | same {same}", flush=True)
# drop-in call latency
for tl in (200, 500, 1000, 2000, 4000):
    t, q = synth_pair(123, tl, 0.05)
    for label, flag in (("score", 0), ("cigar", 1)):
        o = mw.opt_init(flag=flag)
        for _ in range(3): mw.wfa_exact(t, q, o)
        n = 40; t0 = time.perf_counter()
        for _ in range(n): mw.wfa_exact(t, q, o)
        print(f"mwf_wfa_exact {tl} bp {label}: {(time.perf_counter() - t0) / n * 1e6:.1f} us per call", flush=True)
