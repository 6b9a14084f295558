import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
eng = mw.Engine(0)
b = eng.upload(PackedBatch([synth_pair(2001, 150000, 0.035)]))
b.align(mw.opt_init()); print(b.results()[0], eng.stats().kernel_ms)
