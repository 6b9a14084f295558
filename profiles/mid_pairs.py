import sys; sys.path.insert(0,'.')
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
for tl, div in ((4000,0.05),(5000,0.05),(5000,0.07),(6000,0.05)):
    pairs=[synth_pair(4000+i,tl,div) for i in range(4096)]
    pk=PackedBatch(pairs)
    out={}
    for label,blk,pack in (("auto",0,-1),("256x3",256,1),("512x3",512,-1)):
        eng=mw.Engine(0); eng.set("force_kind",2 if blk else -1); eng.set("block",blk); eng.set("band_pack",pack)
        b=eng.upload(pk); o=mw.opt_init()
        for _ in range(3): b.align(o); s,it,nc=b.results()
        st=eng.stats(); out[label]=(round(st.kernel_ms,2), st.block, st.n_retries)
        b.free(); eng.close()
    print(tl,div,out,flush=True)
