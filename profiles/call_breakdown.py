"""Where a single drop-in call spends its time: one 200 bp / 2 kb pair through the engine API, stage by stage
(host buffers -> batch object, enqueue, wait + results, free), next to the kernel time from the library's HIP events
and to the whole mwf_wfa_exact() call."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import miniwfa_amd as mw
from miniwfa_amd.synth import synth_pair, PackedBatch
eng = mw.Engine(0)
for tl in (200, 2000):
    t, q = synth_pair(123, tl, 0.05)
    pk = PackedBatch([(t, q)])
    for flag in (0, 1):
        o = mw.opt_init(flag=flag)
        acc = [0.0] * 4; kms = 0.0; n = 200
        for it in range(n + 10):
            t0 = time.perf_counter(); b = eng.upload(pk)
            t1 = time.perf_counter(); b.align(o)
            t2 = time.perf_counter(); s, _, nc = b.results()
            if flag: b.fetch_cigars()
            t3 = time.perf_counter(); b.free()
            t4 = time.perf_counter()
            if it >= 10:
                acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2; acc[3] += t4 - t3; kms += eng.stats().kernel_ms
        for _ in range(10): mw.wfa_exact(t, q, o)
        t0 = time.perf_counter()
        for _ in range(n): mw.wfa_exact(t, q, o)
        call = (time.perf_counter() - t0) / n * 1e6
        print(f"{tl} bp {'cigar' if flag else 'score'}: s={int(s[0])}  upload {acc[0]/n*1e6:.1f} us | enqueue {acc[1]/n*1e6:.1f} | wait+results {acc[2]/n*1e6:.1f} | free {acc[3]/n*1e6:.1f} "
              f"| kernel (HIP events) {kms/n*1e3:.1f} us -> {kms/n*1e3/max(1,int(s[0])):.2f} us per penalty | mwf_wfa_exact call {call:.1f} us", flush=True)
eng.close()
