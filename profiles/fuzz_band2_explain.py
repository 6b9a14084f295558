"""First penalty at which the device's band of one fuzz pair (profiles/fuzz_band2_oracle.py <seed>, pair index) leaves the oracle's.
Usage: python profiles/fuzz_band2_explain.py <seed> <pair> [block]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
seed, idx = int(sys.argv[1]), int(sys.argv[2])
block = int(sys.argv[3]) if len(sys.argv) > 3 else 512
from miniwfa_amd.synth import fuzz_pairs
t, q = fuzz_pairs(seed, 240)[idx]
import miniwfa_amd as mw
from miniwfa_amd.synth import PackedBatch
from oracle.pyoracle import Oracle, make_opt
orc = Oracle()
o = make_opt()
es, eit, _ = orc.align(t, q, o)
ref = np.array(orc.band_trace(t, q, o), dtype=np.int32).reshape(-1, 2)
eng = mw.Engine(0)
eng.set("force_kind", 2); eng.set("block", block); eng.set("band_pack", 1)
b = eng.upload(PackedBatch([(t, q)]))
go = mw.opt_init()
b.align(go); s, it, nc = b.results()
dev = b.debug_band(go, 0) - 1 - len(t)
n = min(len(dev), len(ref))
bad = np.nonzero((dev[:n] != ref[:n]).any(axis=1))[0]
print("tl ql", len(t), len(q), "device s/n_iter", int(s[0]), int(it[0]), "oracle", es, eit, "band rows", len(dev), len(ref))
if len(bad):
    j = int(bad[0])
    for jj in range(max(0, j - 3), min(n, j + 3)):
        print("penalty", jj + 1, "device", dev[jj].tolist(), "oracle", ref[jj].tolist(), "<-- first divergence" if jj == j else "")
else:
    print("bands agree for", n, "penalties")
open("/tmp/fuzz_pair.txt", "w").write(t.decode() + "\n" + q.decode() + "\n")
