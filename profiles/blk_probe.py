"""The block form of the packed band kernel (band_blk = 1: four penalties per barrier, mwf_band2.hip band4_pass) against the per-penalty form:
(1) correctness on shapes that stress the window (skewed / unrelated / fuzz pairs, 10 kb pairs) against the oracle and against band_blk = 0;
(2) the headline batch (1024 x 10 kb @ 5 %) on four seeds: kernel ms per align.   Usage: python profiles/blk_probe.py [check|time|all]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch, skewed_pairs, fuzz_pairs
what = sys.argv[1] if len(sys.argv) > 1 else "all"

def run(pk, blk, kw=None, tun=()):
    eng = mw.Engine(0); eng.set("band_blk", blk)
    for k, v in tun: eng.set(k, v)
    b = eng.upload(pk); o = mw.opt_init(**(kw or {}))
    b.align(o); s, it, _ = b.results(); st = eng.stats()
    out = (np.array(s).copy(), np.array(it).copy(), st.n_retries, st.kernel_ms, st.block, st.kernel_kind)
    b.free(); eng.close()
    return out

if what in ("check", "all"):
    import fuzzlib as F
    from oracle.pyoracle import Oracle, make_opt
    orc = Oracle()
    sets = {"skewed 2-3 kb": [p for i, p in enumerate(skewed_pairs(1, 400, 200, 3000)) if max(len(p[0]), len(p[1])) >= 1500],
            "skewed 4-9 kb": skewed_pairs(8, 48, 4000, 9000),
            "fuzz 4 kb": [p for p in fuzz_pairs(3, 300, 4000) if len(p[0]) + len(p[1]) > 1500],
            "10 kb @ 5 %": [synth_pair(50000 + i, 10000, 0.05) for i in range(96)],
            "6 kb @ 15 %": [synth_pair(51000 + i, 6000, 0.15) for i in range(64)]}
    bad = 0
    for name, pairs in sets.items():
        pk = PackedBatch(pairs)
        for kw in (dict(), dict(x=6, o1=6, e1=2, o2=30, e2=1), dict(max_s=700)):
            exp = F.oracle_many(orc, pairs, make_opt(**kw))
            for tun in ((("force_kind", 2), ("block", 512), ("band_pack", 1)), (("wide_slots", 4),), ()):
                a = run(pk, 1, kw, tun)
                n_bad = sum(1 for i, e in enumerate(exp) if (int(a[0][i]), int(a[1][i])) != (e[0], e[1]))
                bad += n_bad
                first = [(i, len(pairs[i][0]), len(pairs[i][1]), int(a[0][i]), int(a[1][i]), exp[i][0], exp[i][1]) for i, e in enumerate(exp) if (int(a[0][i]), int(a[1][i])) != (e[0], e[1])][:3]
                print(f"{name} {kw} {tun}: {pk.n} pairs, mismatches vs oracle {n_bad} {first}, re-runs {a[2]}, block {a[4]}", flush=True)
    print("BLK CHECK", "FAILED" if bad else "OK")

if what in ("time", "all"):
    for seed in (50000, 60000, 70000, 80000):
        pk = PackedBatch([synth_pair(seed + i, 10000, 0.05) for i in range(1024)])
        res = {}
        for blk in (0, 1):
            eng = mw.Engine(0); eng.set("band_blk", blk)
            b = eng.upload(pk); o = mw.opt_init()
            ms = []
            for _ in range(6):
                b.align(o); s, it, _ = b.results(); ms.append(eng.stats().kernel_ms)
            res[blk] = (np.median(ms[2:]), int(s.astype(np.int64).sum()), int(it.astype(np.int64).sum()), eng.stats().n_retries)
            b.free(); eng.close()
        print(f"seed {seed}: per-penalty form {res[0][0]:.2f} ms, block form {res[1][0]:.2f} ms ({100 * (res[1][0] / res[0][0] - 1):+.1f} %), same answers {res[0][1:3] == res[1][1:3]}, re-runs {res[0][3]} / {res[1][3]}", flush=True)
