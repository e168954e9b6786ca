#!/usr/bin/env python3
"""Per-kernel counter averages of the decode step from rocprofv3 --pmc passes of tools/pmc_decode.py -> a text table.
    python tools/pmc_decode_counters.py <counter_collection.csv> [<second pass csv> ...]
Derived columns (when the counters are present): matrix-pipe busy share = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8
XCDs); share of wave-cycles spent waiting on any instruction's data = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES; L2 hit rate."""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "dec_" not in n:
            continue
        k = n.replace("void ", "").split("mellow::", 1)[-1].split("(")[0][:48]       # kernel name with its template arguments
        a = acc[k][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for k, cs in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", [0, 0])[1]):
    avg = {c: v / n for c, (n, v) in cs.items()}
    n = max(v[0] for v in cs.values())
    line = f"{k:48s} n={n:4d}"
    if "GRBM_GUI_ACTIVE" in avg:
        cyc = avg["GRBM_GUI_ACTIVE"] / 8.0
        line += f"  cycles/XCD {cyc:8.0f}"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
            line += f"  mfma-busy {avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cyc):6.3f}"
        if "SQ_BUSY_CU_CYCLES" in avg:
            line += f"  cu-busy {avg['SQ_BUSY_CU_CYCLES'] / (256.0 * cyc) :6.3f}"
    if "SQ_WAIT_INST_ANY" in avg and "SQ_WAVE_CYCLES" in avg and avg["SQ_WAVE_CYCLES"]:
        line += f"  waiting {avg['SQ_WAIT_INST_ANY'] / avg['SQ_WAVE_CYCLES']:6.3f}"
    if "SQ_INSTS_VALU" in avg:
        line += f"  valu-insts {avg['SQ_INSTS_VALU']:10.0f}"
    hit, miss = avg.get("TCC_HIT_sum"), avg.get("TCC_MISS_sum")
    if hit is not None and miss is not None and hit + miss > 0:
        line += f"  L2 hit {hit / (hit + miss):6.3f} ({hit + miss:9.0f} requests)"
    print(line)
