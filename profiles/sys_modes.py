"""One long pair on the whole-device kernel in one mode, second call timed (for rocprofv3 / PMC passes: profiles/pmc_cmd.sh <tag> python profiles/sys_modes.py <pair> <mode>).
pair: c4 | mhc; mode: score | cigar | lowmem"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
pair, mode = sys.argv[1], sys.argv[2]
t, q = synth_pair(2001, 150000, 0.035) if pair == "c4" else synth_pair(2002, 5000000, 0.008, 3, 15000)
kw = {"score": {}, "cigar": {"flag": 1}, "lowmem": {"flag": 1, "step": 5000}}[mode]
eng = mw.Engine(0)
b = eng.upload(PackedBatch([(t, q)]))
for rep in range(2):
    b.align(mw.opt_init(**kw)); s, it, nc = b.results()
st = eng.stats()
print(f"{pair} {mode}: s {int(s[0])} n_iter {int(it[0])} cells_pass1 {st.cells_pass1} kernel {st.kernel_ms:.2f} ms peak_device_bytes {st.dev_bytes_peak}", flush=True)
b.free(); eng.close()
