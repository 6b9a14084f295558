"""Per-phase cycle counts of the balanced band kernel (library built with -DMWF_B3_TIMING: profiles/build_variant.sh b3t -DMWF_B3_TIMING,
run with MWF_HIP_LIB=profiles/_b3t_libmwf_hip.so): one 10 kb pair alone, then the same pair with 511 co-resident neighbours."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.api import lib
import ctypes as C
from miniwfa_amd.synth import synth_pair, PackedBatch
eng = mw.Engine(0)
eng.set("band3", 1)
for n in (1, 512)[: (1 if os.environ.get("ONE") else 2)]:
    b = eng.upload(PackedBatch([synth_pair(50000, 10000, 0.05)] * n))
    o = mw.opt_init()
    b.align(o); s = b.results()[0]
    cap = int(s[0])
    buf = np.zeros(2 * cap, dtype=np.uint32)
    got = lib().mwf_gpu_debug_band(eng.h, b.h, C.byref(o), 0, buf.ctypes.data, cap)
    a = buf[0:2 * got:2]; c = buf[1:2 * got:2]
    ph = np.stack([a & 0xffff, a >> 16, c & 0xffff, c >> 16], axis=1).astype(np.float64)
    print(f"pairs {n}: s {int(s[0])}, kernel {eng.stats().kernel_ms:.3f} ms; mean cycles per penalty: header {ph[:,0].mean():.0f}, chunks {ph[:,1].mean():.0f}, drain {ph[:,2].mean():.0f}, barrier+flags {ph[:,3].mean():.0f}, sum {ph.sum(axis=1).mean():.0f}")
    for lo_, hi_ in ((0, 200), (1000, 1200), (2200, 2400)):
        q = ph[lo_:hi_]
        print(f"   penalties {lo_}-{hi_}: {q[:,0].mean():.0f} {q[:,1].mean():.0f} {q[:,2].mean():.0f} {q[:,3].mean():.0f}")
    b.free()
