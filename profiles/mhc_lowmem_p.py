"""The 5 Mb pair in low-memory mode (step 5000, two-pass form) with a chosen block length of the whole-device kernel (sys_p; builds with -DMWF_SYS_ALL_P hold 4, 8 and 16).
Usage: MWF_HIP_LIB=profiles/_allp_libmwf_hip.so python profiles/mhc_lowmem_p.py 8 16"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
gold = {}
for line in open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "long_pairs.jsonl")):
    v = json.loads(line); gold[v["id"]] = v
t, q = synth_pair(2002, 5000000, 0.008, 3, 15000)
for p in [int(x) for x in sys.argv[1:]] or [8]:
    eng = mw.Engine(0)
    eng.set("sys_p", p % 100)
    if p >= 100:
        eng.set("sys_p2", p // 100)   # (e.g. 1608: first pass 8, second pass 16)
    b = eng.upload(PackedBatch([(t, q)])); o = mw.opt_init(flag=1, step=5000)
    for _ in range(2):
        t0 = time.perf_counter(); b.align(o); s, it, nc = b.results(); w = time.perf_counter() - t0
    st = eng.stats(); g = gold["mhc-lowmem"]["expect"]
    ok = (int(s[0]), int(it[0]), int(nc[0])) == (g["s"], g["n_iter"], g["n_cigar"])
    print(f"sys_p {p}: two_pass {st.lowmem_two_pass} kernel {st.kernel_ms:.1f} ms wall {w * 1e3:.1f} ms peak {st.dev_bytes_peak / 1e9:.2f} GB matches golden {ok}", flush=True)
    b.free(); eng.close()
