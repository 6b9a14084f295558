"""The one-diagonal-per-lane kernel (mwf_lane.hip): fuzz against the oracle, then what it buys — single calls and batches of short
pairs with the kernel on (tunable lane_max_len, default 320) and off (0).  python profiles/lane_kernel_probe.py [seeds]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch  # noqa: F401  (one HIP runtime per process: torch's first)
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, fuzz_pairs, PackedBatch
from oracle.pyoracle import Oracle, make_opt

orc = Oracle()
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rng = np.random.default_rng(5)
bad = 0
for seed in range(n_seeds):
    pairs = fuzz_pairs(100 + seed, 150, 400 if seed % 2 else 320)
    pairs += [(b"", b""), (b"A", b""), (b"", b"ACGT"), (b"ACGT", b"ACGT"), (b"ACGTNNRYACGT" * 9, b"ACGTNNRYACGA" * 9), (b"A" * 300, b"A" * 290)]
    for _ in range(30):  # read-like: 5 % divergence, 100-300 bp
        pairs.append(synth_pair(int(rng.integers(1 << 30)), int(rng.integers(100, 400)), 0.05))
    pk = PackedBatch(pairs)
    for kw in (dict(), dict(flag=1), dict(flag=1, o2=4, e2=2), dict(x=2, o1=3, e1=1, o2=6, e2=1), dict(flag=1, x=6, o1=5, e1=3, o2=20, e2=2), dict(flag=1, max_s=20)):
        o = make_opt(**kw)
        eng = mw.Engine(0)
        b = eng.upload(pk)
        b.align(mw.opt_init(**kw))
        s, it, nc = b.results()
        st = eng.stats()
        for i, (t, q) in enumerate(pairs):
            es, eit, ecig = orc.align(t, q, o)
            ok = (int(s[i]), int(it[i])) == (es, eit)
            if ok and ecig is not None and es >= 0:
                ok = b.cigar(i, int(nc[i])).tolist() == ecig
            if not ok:
                bad += 1
                if bad < 10:
                    print("MISMATCH", seed, kw, i, len(t), len(q), (int(s[i]), int(it[i])), (es, eit))
        print("seed", seed, kw, "retries", st.n_retries, "last launch: block", st.block, "packed", st.packed, flush=True)
        b.free(); eng.close()
print("fuzz mismatches:", bad)

def call_us(t, q, o, n=200):
    for _ in range(5): mw.wfa_exact(t, q, o)
    t0 = time.perf_counter()
    for _ in range(n): mw.wfa_exact(t, q, o)
    return (time.perf_counter() - t0) / n * 1e6

def batch_ms(pairs, kw, lane, chunks=2):
    eng = mw.Engine(0)
    eng.set("lane_max_len", lane)
    eng.set("lane_chunks", chunks)
    b = eng.upload(PackedBatch(pairs))
    o = mw.opt_init(**kw)
    for _ in range(2): b.align(o); b.results()
    t0 = time.perf_counter()
    for _ in range(5): b.align(o); r = b.results()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    st = eng.stats()
    b.free(); eng.close()
    return ms, st.n_retries, r[0], st.kernel_ms

for ln in (100, 150, 200, 250, 300):
    t, q = synth_pair(123, ln, 0.05)
    print("single call %d bp: score %.1f us, cigar %.1f us" % (ln, call_us(t, q, mw.opt_init()), call_us(t, q, mw.opt_init(flag=1))), flush=True)
for n, ln, div in ((40000, 150, 0.05), (20000, 200, 0.05), (20000, 250, 0.05), (20000, 300, 0.05), (20000, 300, 0.02), (20000, 150, 0.10)):
    pairs = [synth_pair(7000 + i, ln, div) for i in range(n)]
    bp = sum(len(t) + len(q) for t, q in pairs)
    for kw in (dict(), dict(flag=1)):
        m0, r0, s0, k0 = batch_ms(pairs, kw, 0)
        line = "%d x %d bp @ %.0f %% %s: band kernels %.2f ms wall, %.2f ms kernel (%.2f Gbp/s)" % (n, ln, div * 100, "cigar" if kw else "score", m0, k0, bp / k0 / 1e6)
        for ch in (1, 2, 3, 4):
            m1, r1, s1, k1 = batch_ms(pairs, kw, 320, ch)
            assert (s1 == s0).all()
            line += " | %d chunk(s): %.2f wall, %.2f first launch, %d re-run" % (ch, m1, k1, r1)
        print(line, flush=True)
