#!/usr/bin/env python3
"""Condense a profiles/run_profile.sh output directory (rocprofv3 sqlite output): kernel stats and per-launch PMC means."""
import glob
import os
import sqlite3
import sys

root = sys.argv[1]
for f in sorted(glob.glob(os.path.join(root, "trace", "*.db"))):
    cur = sqlite3.connect(f).cursor()
    print("== kernel stats (rocprofv3 --kernel-trace --stats):", os.path.relpath(f, root))
    print("name | calls | total_ns | avg_ns | pct")
    for r in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        print(" | ".join(str(x) for x in r))
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "*.db")):
        cur = sqlite3.connect(f).cursor()
        rows = list(cur.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"))
        for k, c, v, n in rows:
            if "wfa" in k:
                print(f"== pmc {c} = {v / max(1, n):.6g} per launch ({n} launches) [{k[:70]}]")
# the tree these counters were collected on (the snapshot on the GPU box): fingerprints of the kernels' source files, which
# profiles/make_traffic.py stores in traffic.json and bench.py compares with the tree it times (roofline.frac_stale)
try:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from miniwfa_amd.build import KERNEL_FILES, kernel_fingerprint
    for name in sorted(KERNEL_FILES):
        print(f"== fingerprint {name} {kernel_fingerprint(name)}")
except Exception as e:  # pragma: no cover
    print("== fingerprint unavailable:", e)
