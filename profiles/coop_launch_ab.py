"""Whole-device kernel through hipLaunchCooperativeKernel ("coop_launch" 1, the default) against a plain launch (0): kernel
milliseconds of the 150 kb and 5 Mb pairs (second call of each), results compared."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_golden, golden_inputs
gold = {v["id"]: v for v in load_golden("long_pairs.jsonl")} if os.path.exists(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "long_pairs.jsonl")) else {}
cases = [(k, v) for k, v in gold.items()]
for gid, v in cases:
    t, q = golden_inputs(v)
    kw = {k: v["opt"][k] for k in ("flag", "step")}
    for mode in (1, 0, 1, 0):
        eng = mw.Engine(0); eng.set("coop_launch", mode)
        b = eng.upload(PackedBatch([(t, q)]))
        for rep in range(2):
            b.align(mw.opt_init(**kw)); s, it, nc = b.results()
        st = eng.stats()
        ok = (int(s[0]), int(it[0])) == (v["expect"]["s"], v["expect"]["n_iter"])
        print(f"{gid} tl {len(t)} {kw} coop_launch {mode}: kernel {st.kernel_ms:.2f} ms, kind {st.kernel_kind}, equals the reference: {ok}", flush=True)
        b.free(); eng.close()
