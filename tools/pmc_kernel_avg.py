#!/usr/bin/env python3
"""Developer tool (CPU): per-kernel averages of the counters in a `rocprofv3 --pmc ... --output-format csv` counter_collection CSV.
    python tools/pmc_kernel_avg.py <counter_collection.csv> [kernel-name substring ...]"""
import collections
import csv
import sys

rows = csv.DictReader(open(sys.argv[1]))
want = sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in rows:
    k = r["Kernel_Name"]
    if want and not any(w in k for w in want):
        continue
    a = acc[k.split("(")[0][-60:]][r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
for k, cs in acc.items():
    print(k)
    for c, (n, tot) in sorted(cs.items()):
        print(f"    {c:44s} n={n:6d} avg={tot / n:16.1f}")
