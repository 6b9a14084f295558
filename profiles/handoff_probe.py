"""How long does a published edge flag take to be seen?  (library built with -DMWF_BAND_TIMING: the whole-device kernel
then leaves, per penalty, four s_memrealtime stamps (100 MHz) in the band-trace buffer: low edge published by its owner,
workgroup 5 starts looking, sees the low-edge flag, sees all flags.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
eng = mw.Engine(0)
eng.set("force_kind", 1)
b = eng.upload(PackedBatch([synth_pair(2001, 150000, 0.035)]))
raw = b.debug_band(mw.opt_init(), 0, cap=1 << 16)            # (n, 2) int32 = n x 8 bytes
u = raw.reshape(-1).view(np.uint64)
t = u[:len(u) // 4 * 4].reshape(-1, 4)[2:]                    # one row of 4 stamps per penalty (the first quarter of the run)
t = t[(t != 0).all(axis=1)]
ns = 10.0
pub, look, see_lo, see_all = (t[:, i].astype(np.int64) for i in range(4))
period = np.diff(see_all)
print("penalties with all four stamps:", len(t))
for name, v in (("period (all flags seen -> all flags seen)", period), ("published -> seen (low edge)", see_lo - pub), ("published -> all seen", see_all - pub),
                ("start looking -> all seen", see_all - look), ("all seen (previous) -> published", pub[1:] - see_all[:-1])):
    v = v * ns
    print(f"{name}: median {np.median(v):.0f} ns, p10 {np.percentile(v, 10):.0f}, p90 {np.percentile(v, 90):.0f}")
