"""Cycle counts of the block form (library built with -DMWF_B4_TIMING: profiles/build_band2_variant.sh b4t -DMWF_BAND_DEV -DMWF_B4_TIMING; run with
MWF_HIP_LIB=profiles/_b4t_libmwf_hip.so): one 10 kb pair alone, per wave: cycles per block of four penalties, split."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ctypes as C
import miniwfa_amd as mw
from miniwfa_amd.api import lib
from miniwfa_amd.synth import synth_pair, PackedBatch
eng = mw.Engine(0); eng.set("band_blk", 1)
for n in (1, 512):
    b = eng.upload(PackedBatch([synth_pair(50000, 10000, 0.05)] * n))
    o = mw.opt_init()
    b.align(o); s = b.results()[0]; cap = int(s[0])
    print(f"pairs {n}: s {cap}, kernel {eng.stats().kernel_ms:.3f} ms")
    for wave in range(8):
        o2 = mw.opt_init(max_iter=-(64 * wave) if wave else 0)
        buf = np.zeros(2 * cap, dtype=np.int32)
        got = lib().mwf_gpu_debug_band(eng.h, b.h, C.byref(o2), 0, buf.ctypes.data, cap)
        r = buf[: 2 * (got // 4) * 4].reshape(-1, 8)
        r = r[1:-1]
        tot, ref, cmp_, bar, nrun, hdr = r[:, 0], r[:, 1], r[:, 2], r[:, 3], r[:, 4], r[:, 5]
        print(f"  wave {wave}: blocks {len(r)}; per block: total (barrier to pre-barrier) {tot.mean():.0f}, bookkeeping+header {hdr.mean():.0f}, refresh {ref.mean():.0f}, compute {cmp_.mean():.0f} ({nrun.mean():.2f} chunks -> {cmp_.sum() / max(1, nrun.sum()) / 4:.0f} per chunk-penalty), barrier wait {bar.mean():.0f}")
        for k in (1, 2, 3):
            q = r[nrun == k]
            if len(q): print(f"      blocks with {k} chunk(s) ({len(q)}): total {q[:,0].mean():.0f}, header {q[:,5].mean():.0f}, refresh {q[:,1].mean():.0f}, compute {q[:,2].mean():.0f}, barrier wait {q[:,3].mean():.0f}")
    b.free()
