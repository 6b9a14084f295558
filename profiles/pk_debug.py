"""Development check of the packed band kernel: small goldens one option group at a time, errors printed (no pytest capture)."""
import sys, os, faulthandler
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
from conftest import load_golden, golden_inputs
from oracle.pyoracle import Oracle, make_opt, cigar_str as ocig
OPT_KEYS = ("flag", "x", "o1", "e1", "o2", "e2", "step", "max_s", "max_iter")
eng = mw.Engine(0)
vecs = [v for v in load_golden("exact_small.jsonl") if v["entry"] == "exact"]
if len(sys.argv) > 1: vecs = vecs[:int(sys.argv[1])]
groups = {}
for v in vecs:
    groups.setdefault(tuple(v["opt"][k] for k in OPT_KEYS), []).append(v)
nbad = 0
for key, vs in groups.items():
    o = mw.opt_init(**dict(zip(OPT_KEYS, key)))
    pairs = [golden_inputs(v) for v in vs]
    print("group", key, len(vs), "pairs", flush=True)
    b = eng.upload(PackedBatch(pairs))
    b.align(o)
    s, it, nc = b.results()
    st = eng.stats()
    for i, v in enumerate(vs):
        exp = v["expect"]
        ok = s[i] == exp["s"] and it[i] == exp["n_iter"]
        if ok and exp["cigar"] is not None: ok = ocig(b.cigar(i, int(nc[i]))) == exp["cigar"]
        if not ok:
            nbad += 1
            if nbad < 20: print("  BAD", v["id"], "tl/ql", len(pairs[i][0]), len(pairs[i][1]), "got", int(s[i]), int(it[i]), "exp", exp["s"], exp["n_iter"], "block", st.block, flush=True)
    b.free()
print("bad:", nbad)
