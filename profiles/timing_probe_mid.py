"""Cycles per penalty inside the mid kernel (library built with -DMWF_MID_TIMING by profiles/build_variant.sh, run with MWF_HIP_LIB=...):
header | groups (recurrence + extension) | flags .. barrier | flag read, for three waves of the workgroup."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.api import lib
import ctypes as C
from miniwfa_amd.synth import synth_pair, PackedBatch
for tl, blk in ((500, 512), (1000, 512), (2000, 1024), (2000, 512)):
    eng = mw.Engine(0); eng.set("mid_block", blk)
    b = eng.upload(PackedBatch([synth_pair(4000, tl, 0.05)]))
    o = mw.opt_init()
    b.align(o); s = b.results()[0]
    cap = int(s[0])
    for wave in (0, 1, blk // 64 - 1):
        o2 = mw.opt_init(max_iter=-(64 * wave) if wave else 0)
        buf = np.zeros(2 * cap, dtype=np.uint32)
        got = lib().mwf_gpu_debug_band(eng.h, b.h, C.byref(o2), 0, buf.ctypes.data, cap)
        a = buf[0:2 * got:2]; c = buf[1:2 * got:2]
        ph = np.stack([a & 0xffff, a >> 16, c & 0xfff, (c >> 12) & 0xffff, c >> 28], axis=1).astype(np.float64)[8:]
        print(f"{tl} bp, {blk} threads, wave {wave}: penalties {len(ph)}: header {ph[:,0].mean():.0f}, groups {ph[:,1].mean():.0f} ({ph[:,4].mean():.2f} per penalty), flags..barrier {ph[:,2].mean():.0f}, flag read {ph[:,3].mean():.0f}, sum {ph[:,:4].sum(axis=1).mean():.0f}", flush=True)
    b.free(); eng.close()
