"""Per-phase cycle counts of the packed band kernel (library built with -DMWF_B2_TIMING: profiles/build_variant.sh b2t -DMWF_BAND_DEV -DMWF_B2_TIMING,
run with MWF_HIP_LIB=profiles/_b2t_libmwf_hip.so): one 10 kb pair alone and with 511 co-resident neighbours, as seen by wave 0 and by
the wave that holds the window's first chunk late in the pair."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.api import lib
import ctypes as C
from miniwfa_amd.synth import synth_pair, PackedBatch
eng = mw.Engine(0)
for n in (1, 512):
    b = eng.upload(PackedBatch([synth_pair(50000, 10000, 0.05)] * n))
    o = mw.opt_init()
    b.align(o); s = b.results()[0]
    cap = int(s[0])
    for wave in range(8):
        o2 = mw.opt_init(max_iter=-(64 * wave) if wave else 0)
        buf = np.zeros(2 * cap, dtype=np.uint32)
        got = lib().mwf_gpu_debug_band(eng.h, b.h, C.byref(o2), 0, buf.ctypes.data, cap)
        a = buf[0:2 * got:2]; c = buf[1:2 * got:2]
        ph = np.stack([a & 0xffff, a >> 16, c & 0xfff, (c >> 12) & 0xffff, c >> 28], axis=1).astype(np.float64)
        tot = ph[:, :4].sum(axis=1)
        print(f"pairs {n} wave {wave}: penalties {got}, kernel {eng.stats().kernel_ms:.3f} ms; mean cycles per penalty: header {ph[:,0].mean():.0f}, chunks {ph[:,1].mean():.0f} ({ph[:,4].mean():.2f} chunks), drain {ph[:,2].mean():.0f}, barrier+flags {ph[:,3].mean():.0f}, sum {tot.mean():.0f}")
        for k in (0, 1, 2, 3):
            q = ph[ph[:, 4] == k]
            if len(q): print(f"      penalties with {k} chunk(s) on this wave ({len(q)}): header {q[:,0].mean():.0f}, chunks {q[:,1].mean():.0f}, drain {q[:,2].mean():.0f}, barrier+flags {q[:,3].mean():.0f}")
    b.free()
