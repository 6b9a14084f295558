"""Whole-device kernel with and without the deferred match extension (MWF_SYS_NODEFER: bit 0 provenance pass, bit 1 traceback passes), by mode.
Usage: [MWF_SYS_NODEFER=n] python profiles/sys_defer_probe.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
for name, args, modes in (("c4", (2001, 150000, 0.035), (("cigar", dict(flag=1), 0), ("lowmem two-pass", dict(flag=1, step=5000), 1))),
                          ("mhc", (2002, 5000000, 0.008, 3, 15000), (("lowmem two-pass", dict(flag=1, step=5000), 0), ("cigar high-memory", dict(flag=1), 0)))):
    t, q = synth_pair(*args)
    for label, kw, budget in modes:
        eng = mw.Engine(0)
        if budget: eng.set("lowmem_budget_mb", budget)
        b = eng.upload(PackedBatch([(t, q)]))
        for _ in range(2):
            b.align(mw.opt_init(**kw)); s, it, nc = b.results()
        st = eng.stats()
        print(f"{name} {label}: s {int(s[0])} n_cigar {int(nc[0])} kernel {st.kernel_ms:.1f} ms two_pass {st.lowmem_two_pass} peak {st.dev_bytes_peak / 1e9:.2f} GB", flush=True)
        b.free(); eng.close()
