import sys, time
sys.path.insert(0,'/root/repo')
import numpy as np, miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
pairs=[synth_pair(50000+i,10000,0.05) for i in range(1024)]
eng=mw.Engine(0)
for n in (1,8,64,256,512,1024):
    b=eng.upload(PackedBatch(pairs[:n])); o=mw.opt_init()
    for _ in range(2): b.align(o); s,it,nc=b.results()
    ms=[]
    for _ in range(3):
        b.align(o); s,it,nc=b.results(); ms.append(eng.stats().kernel_ms)
    steps=float(s.max()); print(f"n={n} kernel_ms={min(ms):.3f} max_s={steps} us/step(if serial per WG)={min(ms)*1e3/ (steps*max(1,n/256)):.3f} grid={eng.stats().grid} block={eng.stats().block}")
    b.free()
