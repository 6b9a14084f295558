#!/usr/bin/env python3
"""Condense a profiles/run_profile.sh output directory: kernel stats and per-launch PMC sums."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats", f)
    for row in csv.DictReader(open(f)):
        print({k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(set)
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "?")
            if "wfa" not in k:
                continue
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k].add(row.get("Dispatch_Id"))
        for k in acc:
            n = max(1, len(cnt[k]))
            print("== pmc", os.path.basename(d), k[:60], "launches", n, {c: v / n for c, v in acc[k].items()})
