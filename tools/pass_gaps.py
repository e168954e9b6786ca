#!/usr/bin/env python3
"""Developer tool (CPU): where a generate() pass leaves the GPU idle.  From a `rocprofv3 --kernel-trace` CSV of
tools/decode_probe.py (or bench.py): the passes are cut at the reflect-pad launch that opens the encoder; for the last pass the
tool prints every idle interval of at least MIN_US between the end of the latest-ending kernel so far and the next kernel start
(with the kernels on both sides), and the total idle time by phase.  usage: pass_gaps.py <kernel_trace.csv> [min_us=6]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "reflect_pad_kernel" in r["Kernel_Name"]]
if len(starts) < 2:
    sys.exit("need at least two passes in the trace")
a, b = starts[-2], starts[-1]            # the last COMPLETE pass
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])
end_last = max(int(r["End_Timestamp"]) for r in seg)
print(f"pass: {len(seg)} launches, {(end_last - t0) / 1e3:.1f} us from its first kernel start to its last kernel end; "
      f"next pass starts {(int(rows[b]['Start_Timestamp']) - end_last) / 1e3:.1f} us later")
busy_until = int(seg[0]["End_Timestamp"])
prev = seg[0]
tot = {"small (< min)": 0.0, "listed": 0.0}
for r in seg[1:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > busy_until:
        g = (s - busy_until) / 1e3
        if g >= min_us:
            tot["listed"] += g
            print(f"  +{(busy_until - t0) / 1e3:9.1f} us  idle {g:7.1f} us   after {prev['Kernel_Name'][:48]:48s} before {r['Kernel_Name'][:48]}")
        else:
            tot["small (< min)"] += g
    if e > busy_until:
        busy_until, prev = e, r
print(f"idle listed {tot['listed']:.1f} us; idle in gaps below {min_us} us (launch boundaries) {tot['small (< min)']:.1f} us")
