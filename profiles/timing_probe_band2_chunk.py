"""Cycle counts inside the first active chunk of a wave (library built with -DMWF_BAND_DEV -DMWF_B2_TIMING=2 by profiles/build_variant.sh,
run with MWF_HIP_LIB=...): rows + recurrence | masks, liveness, geometry, edge stores | first probe | walks + store."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.api import lib
import ctypes as C
from miniwfa_amd.synth import synth_pair, PackedBatch
eng = mw.Engine(0)
if len(sys.argv) > 1: eng.set("band_fold", int(sys.argv[1]))
for n in (1, 512):
    b = eng.upload(PackedBatch([synth_pair(50000, 10000, 0.05)] * n))
    o = mw.opt_init()
    b.align(o); s = b.results()[0]
    cap = int(s[0])
    for wave in (0, 3, 7):
        o2 = mw.opt_init(max_iter=-(64 * wave) if wave else 0)
        buf = np.zeros(2 * cap, dtype=np.uint32)
        got = lib().mwf_gpu_debug_band(eng.h, b.h, C.byref(o2), 0, buf.ctypes.data, cap)
        a = buf[0:2 * got:2]; c = buf[1:2 * got:2]
        ph = np.stack([a & 0xfff, (a >> 12) & 0xfff, c & 0xfff, (c >> 12) & 0xfff, a >> 28], axis=1).astype(np.float64)
        q = ph[ph[:, 4] >= 1]
        print(f"pairs {n} wave {wave}: penalties with an active chunk {len(q)}: rows+recurrence {q[:,0].mean():.0f}, masks/liveness/geometry/edge stores {q[:,1].mean():.0f}, first probe {q[:,2].mean():.0f}, walks+store {q[:,3].mean():.0f}, sum {q[:,:4].sum(axis=1).mean():.0f}")
    b.free()
