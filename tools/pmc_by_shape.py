#!/usr/bin/env python3
"""Developer tool (CPU): memory-side traffic per (kernel, grid) from two rocprofv3 counter CSVs (one --pmc FETCH_SIZE pass, one
--pmc WRITE_SIZE pass of the same workload; FETCH_SIZE x2 as in tools/pmc_traffic.py).  usage: pmc_by_shape.py fetch.csv write.csv"""
import collections
import csv
import sys


def load(path, counter):
    per = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = (r["Kernel_Name"][:44], r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", "?"))
        per[k][0] += 1
        per[k][1] += float(r["Counter_Value"])
    return per


f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
rows = []
for k in f:
    n = f[k][0]
    rd, wr = 2.0 * f[k][1] * 1024 / n, w.get(k, [1, 0.0])[1] * 1024 / max(1, w.get(k, [1, 0.0])[0])
    rows.append((n * (rd + wr), k, n, rd, wr))
for tot, k, n, rd, wr in sorted(rows, reverse=True)[:30]:
    print(f"{k[0]:44s} grid={k[1]:>9s} n={n:5d} read {rd / 1e6:8.1f} MB  write {wr / 1e6:8.1f} MB per launch   total {tot / 1e9:7.2f} GB")
