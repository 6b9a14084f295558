"""Cycles per penalty inside the lane kernel (library built with -DMWF_LANE_TIMING by profiles/build_variant.sh, run with MWF_HIP_LIB=...):
header | chunks | footer of the wave of one traced pair — alone on the device, and as pair 20000 of a batch of 40 000."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.api import lib
import ctypes as C
from miniwfa_amd.synth import synth_pair, PackedBatch
for n, which in ((1, 0), (40000, 20000)):
    eng = mw.Engine(0)
    pairs = [synth_pair(7000 + i, 150, 0.05) for i in range(n)]
    b = eng.upload(PackedBatch(pairs))
    o = mw.opt_init()
    b.align(o); s = b.results()[0]
    cap = int(s[which])
    buf = np.zeros(2 * cap, dtype=np.uint32)
    got = lib().mwf_gpu_debug_band(eng.h, b.h, C.byref(o), which, buf.ctypes.data, cap)
    a = buf[0:2 * got:2]; c = buf[1:2 * got:2]
    ph = np.stack([a & 0xffff, a >> 16, c & 0xffff, (c >> 16) & 0x7fff, c >> 31], axis=1).astype(np.float64)
    print(f"{n} pair(s), pair {which}: s {cap}: header {ph[:,0].mean():.0f}, chunks {ph[:,1].mean():.0f} ({ph[:,3].mean():.2f} per penalty), footer {ph[:,2].mean():.0f}", flush=True)
    b.free(); eng.close()
