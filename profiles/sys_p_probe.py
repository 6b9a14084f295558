"""Block length of the whole-device kernel (penalties per hand-off): 8 (the product) against 16 on the 150 kb pair, every mode — needs a library built with -DMWF_SYS_ALL_P
(profiles/build_variant.sh sysp -DMWF_SYS_ALL_P; MWF_HIP_LIB=profiles/_sysp_libmwf_hip.so)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
pairs = {"c4": synth_pair(2001, 150000, 0.035)}
if len(sys.argv) > 1 and sys.argv[1] == "mhc": pairs["mhc"] = synth_pair(2002, 5000000, 0.008, 3, 15000)
for name, (t, q) in pairs.items():
    for mode, kw in (("score", {}), ("cigar", {"flag": 1}), ("lowmem", {"flag": 1, "step": 5000})):
        res = {}
        for p in (8, 16):
            eng = mw.Engine(0)
            try: eng.set("sys_p", p)
            except ValueError: continue
            b = eng.upload(PackedBatch([(t, q)]))
            for rep in range(2):
                b.align(mw.opt_init(**kw)); s, it, nc = b.results()
            st = eng.stats()
            res[p] = (st.kernel_ms, int(s[0]), int(it[0]), int(nc[0]), st.n_retries)
            b.free(); eng.close()
        print(name, mode, {p: f"{v[0]:.2f} ms (s {v[1]}, retries {v[4]})" for p, v in res.items()}, "same answers", len({v[1:4] for v in res.values()}) == 1, flush=True)
