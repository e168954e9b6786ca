#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) launch count / avg / min / total, and the
busy time of one decode step."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    key = (r["Kernel_Name"][:64], r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"])
    agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print(f"{k[0]:64s} grid=({k[1]},{k[2]}) wg={k[3]:>5s} n={len(v):6d} avg={sum(v) / len(v):9.2f}us min={min(v):8.2f} tot={sum(v) / 1e3:9.2f}ms")
print("total kernel ms", sum(sum(v) for v in agg.values()) / 1e3)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "argmax_cand" in r["Kernel_Name"]]
if len(idx) > 3:
    seg = rows[idx[-3] + 1: idx[-2] + 1]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    print("one decode step: kernels", len(seg), "busy us", busy / 1e3)
