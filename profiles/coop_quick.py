"""Kernel seconds of the whole-device kernel on the C4-like 150 kb pair (score, CIGAR) and the MHC-like 5 Mb pair (score), second call of each.
Usage: python profiles/coop_quick.py [mhc]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
cases = [("c4_like_150kb", synth_pair(2001, 150000, 0.035), [dict(), dict(flag=1)])]
if len(sys.argv) > 1: cases.append(("mhc_like_5Mb", synth_pair(2002, 5000000, 0.008, 3, 15000), [dict()]))
for name, (t, q), modes in cases:
    eng = mw.Engine(0)
    for name_, env in (("sys_p", "MWF_SYS_P"),):
        if os.environ.get(env) is not None:
            eng.set(name_, int(os.environ[env]))
    b = eng.upload(PackedBatch([(t, q)]))
    for kw in modes:
        for rep in range(2):
            b.align(mw.opt_init(**kw)); s, it, nc = b.results()
        print(f"{name} {kw}: s {int(s[0])} n_iter {int(it[0])} kernel {eng.stats().kernel_ms:.2f} ms", flush=True)
    b.free(); eng.close()
