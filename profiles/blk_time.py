"""Kernel ms of the headline batch on the per-penalty and block forms, seeds 50000..80000 (MWF_HIP_LIB picks the library variant)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
seeds = [int(x) for x in sys.argv[1:]] or [50000, 60000, 70000, 80000]
for seed in seeds:
    pk = PackedBatch([synth_pair(seed + i, 10000, 0.05) for i in range(1024)])
    res = {}
    for blk in (0, 1):
        eng = mw.Engine(0); eng.set("band_blk", blk)
        b = eng.upload(pk); o = mw.opt_init()
        ms = []
        for _ in range(6):
            b.align(o); s, it, _ = b.results(); ms.append(eng.stats().kernel_ms)
        res[blk] = (np.median(ms[2:]), int(s.astype(np.int64).sum()), int(it.astype(np.int64).sum()), eng.stats().n_retries)
        b.free(); eng.close()
    print(f"seed {seed}: per-penalty form {res[0][0]:.2f} ms, block form {res[1][0]:.2f} ms ({100 * (res[1][0] / res[0][0] - 1):+.1f} %), same answers {res[0][1:3] == res[1][1:3]}, re-runs {res[0][3]} / {res[1][3]}", flush=True)
