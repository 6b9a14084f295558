"""Where a mwf_wfa_chain call's time goes (MWF_CHAIN_TIMING / MWF_SHARE_TIMING diagnostics of the library), on bench.py's chain_mode pairs.
Usage (GPU box): python profiles/chain_probe.py [reps] [only this target length]"""
import os
import sys
import time

os.environ["MWF_CHAIN_TIMING"] = "1"
os.environ["MWF_SHARE_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (first: the HIP runtime torch ships)
from miniwfa_amd import api as mw
from miniwfa_amd.synth import synth_pair

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
only = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for tl, p in ((5000, 0.05), (30000, 0.04), (100000, 0.03)):
    if only and tl != only:
        continue
    t, q = synth_pair(4242, tl, p, 2, 800)
    o = mw.opt_init(flag=1)
    for _ in range(2):
        mw.wfa_chain(t, q, o)
    sys.stderr.flush()
    print(f"== {tl} bp @ {p}", flush=True)
    t0 = time.perf_counter()
    for _ in range(reps):
        mw.wfa_chain(t, q, o)
    print(f"   {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per call", flush=True)
