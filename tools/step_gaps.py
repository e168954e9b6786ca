#!/usr/bin/env python3
"""Developer tool: from a rocprofv3 --kernel-trace CSV of tools/decode_probe.py, the anatomy of a decode step: period (arg-max end to
arg-max end), sum of kernel durations, and the idle time between the last kernel of a step and the first kernel of the next
(the hipGraph launch boundary)."""
import csv, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "dec_argmax_kernel" in r["Kernel_Name"]]
per, busy, gap, inner = [], [], [], []
for a, b in zip(idx[:-1], idx[1:]):
    seg = rows[a + 1: b + 1]
    if len(seg) < 100 or len(seg) > 140:
        continue                      # a prefill sits in between
    per.append((int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3)
    busy.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e3)
    gap.append((int(seg[0]["Start_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3)
    inner.append(sum(int(y["Start_Timestamp"]) - int(x["End_Timestamp"]) for x, y in zip(seg[:-1], seg[1:])) / 1e3)
print(f"steps {len(per)}: period {statistics.median(per):.1f} us, kernel durations {statistics.median(busy):.1f} us, "
      f"gap before the step's first kernel {statistics.median(gap):.2f} us, gaps between its kernels (sum) {statistics.median(inner):.1f} us")
