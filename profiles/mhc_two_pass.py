"""The 5 Mb pair in low-memory mode (step 5000), walk variant (first pass stores its whole traceback) against the true two-pass form
(provenance + snapshots, reference mwf_wfa_seg): kernel seconds, peak device memory, equality with the golden answer.
Usage: python profiles/mhc_two_pass.py"""
import sys, os, json, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import numpy as np
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
gold = {}
for line in open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "long_pairs.jsonl")):
    v = json.loads(line); gold[v["id"]] = v
for name, args, gid in (("c4", (2001, 150000, 0.035), "c4-lowmem"), ("mhc", (2002, 5000000, 0.008, 3, 15000), "mhc-lowmem")):
    t, q = synth_pair(*args)
    for budget in (0, 1000):
        eng = mw.Engine(0)
        if budget: eng.set("lowmem_budget_mb", budget)
        b = eng.upload(PackedBatch([(t, q)])); o = mw.opt_init(flag=1, step=5000)
        ks = []
        for _ in range(2):
            t0 = time.perf_counter(); b.align(o); s, it, nc = b.results(); w = time.perf_counter() - t0; ks.append(eng.stats().kernel_ms)
        st = eng.stats()
        cg = b.cigar(0, int(nc[0]))
        g = gold[gid]["expect"]
        ok = (int(s[0]), int(it[0]), int(nc[0])) == (g["s"], g["n_iter"], g.get("n_cigar", int(nc[0])))
        print(f"{name} low-memory budget {budget or 'default'}: two_pass {st.lowmem_two_pass} kernel {ks[-1]:.1f} ms wall {w * 1e3:.1f} ms peak {st.dev_bytes_peak / 1e9:.2f} GB cells_pass1 {st.cells_pass1} matches golden {ok} retries {st.n_retries}", flush=True)
        b.free(); eng.close()
