import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
for n in (1, 4, 16, 64):
  for tl in (100, 200, 300, 400):
    pairs = [synth_pair(123 + i, tl, 0.05) for i in range(n)]
    pk = PackedBatch(pairs)
    for flag in (0, 1):
        out = []
        for mode in ("lane", "mid256", "mid512"):
            eng = mw.Engine(0)
            if mode != "lane":
                eng.set("lane_max_len", 0); eng.set("mid_block", int(mode[3:]))
            b = eng.upload(pk); o = mw.opt_init(flag=flag)
            for _ in range(3): b.align(o); b.results()
            ks, ws = [], []
            for _ in range(20):
                t0 = time.perf_counter(); b.align(o); b.results(); ws.append((time.perf_counter() - t0) * 1e6); ks.append(eng.stats().kernel_ms * 1e3)
            st = eng.stats()
            out.append(f"{mode} k {np.median(ks):6.1f} us step {np.median(ws):6.1f} ({st.packed})")
            b.free(); eng.close()
        print(f"{n:3d} x {tl} bp flag {flag}: " + " | ".join(out), flush=True)
