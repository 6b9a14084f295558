import sys, os, time; sys.path.insert(0,'.')
from oracle.pyoracle import Oracle, Reference, make_opt
from miniwfa_amd.synth import synth_pair, PackedBatch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
pk=PackedBatch([synth_pair(50000+i,10000,0.05) for i in range(512)])
o=Oracle(); r=Reference()
for th in (1, 8, 16, 32, 64, 128, 256):
    n = min(512, max(8, th*2))
    s,it,sec=o.batch(pk, make_opt(), th, exact_fn=r.exact_addr(), arena=r.arena_addrs(), n=n)
    print("threads", th, "pairs", n, "sec", round(sec,3), "pairs/s", round(n/sec,1), flush=True)
