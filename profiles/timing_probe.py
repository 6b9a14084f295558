"""Per-phase cycle counts of the band kernel (library built with -DMWF_BAND_TIMING):
one pair alone on the device, then the same pair with a co-resident neighbour workgroup."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
eng = mw.Engine(0)
for n in (1, 512):
    b = eng.upload(PackedBatch([synth_pair(50000, 10000, 0.05)] * n))
    b.align(mw.opt_init()); print(n, b.results()[0][:1], eng.stats().kernel_ms, flush=True)
    b.free()
