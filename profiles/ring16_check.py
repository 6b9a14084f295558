"""Generic kernel with 16-bit ring rows (ring16 = 1, the default) against 32-bit rows (ring16 = 0) on batches that take its E2/F2-in-LDS
path: identical s / n_iter / CIGAR, kernel times of both; plus a pair whose offsets outgrow 16 bits (re-run with 32-bit rows).
Usage: python profiles/ring16_check.py [quick]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch

def run(pk, flag, ring16, reps=2, force_generic=False):
    eng = mw.Engine(0)
    eng.set("ring16", 2 if ring16 else 0)   # 2: also for batches of fewer pairs than CUs
    if force_generic: eng.set("force_kind", 0)
    b = eng.upload(pk)
    o = mw.opt_init(flag=flag)
    for _ in range(reps):
        b.align(o); s, it, nc = b.results()
    st = eng.stats()
    cig = [b.cigar(i, int(nc[i])).tolist() for i in range(pk.n)] if flag else None
    out = (np.array(s), np.array(it), cig, st.kernel_ms, st.kernel_kind, st.block, st.n_retries)
    b.free(); eng.close()
    return out

def compare(name, pairs, flags=(0, 1), force_generic=False):
    pk = PackedBatch(pairs)
    for flag in flags:
        a = run(pk, flag, 0, force_generic=force_generic); c = run(pk, flag, 1, force_generic=force_generic)
        ok = (a[0] == c[0]).all() and (a[1] == c[1]).all() and a[2] == c[2]
        print(f"{name} flag={flag}: 32-bit rows {a[3]:.2f} ms (kind {a[4]} block {a[5]} retries {a[6]}) | 16-bit rows {c[3]:.2f} ms (kind {c[4]} block {c[5]} retries {c[6]}) | identical {ok} | mean s {a[0].mean():.0f}", flush=True)

quick = len(sys.argv) > 1
compare("16 x 20kb @8%", [synth_pair(7000 + i, 20000, 0.08) for i in range(16)])
compare("64 x 50kb @3%", [synth_pair(60000 + i, 50000, 0.03) for i in range(64)])
compare("ragged 13-45kb", [synth_pair(7100 + i, 13000 + 4000 * (i % 9), 0.02 + 0.01 * (i % 5)) for i in range(27)])
compare("52kb @10% (offsets outgrow 16 bits: re-run)", [synth_pair(7200, 52000, 0.10), synth_pair(7201, 30000, 0.03)], flags=(0, 1), force_generic=True)
if not quick:
    compare("1250 x 50kb @3% (one GPU's share of configs[4])", [synth_pair(60000 + i, 50000, 0.03) for i in range(1250)], flags=(0,))
