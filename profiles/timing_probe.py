import sys
sys.path.insert(0,'/root/repo')
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
eng=mw.Engine(0)
b=eng.upload(PackedBatch([synth_pair(50000,10000,0.05)]))
b.align(mw.opt_init()); print(b.results()[0], eng.stats().kernel_ms)
