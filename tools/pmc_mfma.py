#!/usr/bin/env python3
"""MFMA utilisation of the dense GEMM family from a rocprofv3 --pmc pass of tools/pmc_prefill.py
(counters SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE, whichever the pass
collected) -> profiles/rNN_pmc_gemm_mfma.json.   python tools/pmc_mfma.py <counter_collection.csv> <out.json>
SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles summed over the chip's 1024 SIMDs (32 per v_mfma_f32_32x32x16_bf16,
MI355X_MICROARCH.md; checked: gate/up prefill = 2352 tiles x 4 waves x 36 k-steps x 24 MFMAs x 32 = 260.1 M = the counter);
GRBM_GUI_ACTIVE is the kernel's busy shader-clock cycles SUMMED over the 8 XCDs (2.83 M for a 203 us kernel = 8 x 353 k, i.e. an
effective clock of 1.74 GHz under this load).  utilisation = busy / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): the share of the cycles
the chip actually ran in which a SIMD's matrix pipe was busy (the TFLOP/s figure of bench.py is against the 2.4 GHz peak)."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_sha16  # noqa: E402

NAMES = ("gemm_x3q_kernel", "gemm_x3p_kernel", "gemm_x3w_kernel", "gemm_bf16x3f_kernel", "gemm_f32_kernel", "prefill_attention_x3_kernel",
         "prefill_attention_kernel", "window_attention_mfma_kernel")
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = next((n for n in NAMES if n in r["Kernel_Name"]), None)
    if k is None:
        continue
    tag = k + ("<" + r["Kernel_Name"].split("<")[1].split(">")[0] + ">" if "<" in r["Kernel_Name"] else "")
    a = acc[tag][r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
out = {"source_sha16": kernel_source_sha16(), "precision": os.environ.get("MELLOW_PRECISION", "f32x3"),
       "source": "rocprofv3 --pmc (one pass) of tools/pmc_prefill.py; reduced with tools/pmc_mfma.py", "kernels": {}}
tot_busy = tot_act = 0.0
for k, cs in acc.items():
    e = {c: {"launches": n, "avg": v / n} for c, (n, v) in cs.items()}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "GRBM_GUI_ACTIVE" in cs:
        busy, act = cs["SQ_VALU_MFMA_BUSY_CYCLES"][1], cs["GRBM_GUI_ACTIVE"][1]
        e["mfma_utilisation"] = round(busy / (1024.0 * act / 8.0), 4) if act else None
        e["xcd_cycles_per_launch"] = round(act / 8.0 / cs["GRBM_GUI_ACTIVE"][0])
        if k.startswith("gemm_"):
            tot_busy += busy
            tot_act += act
    out["kernels"][k] = e
out["gemm_family_mfma_utilisation"] = round(tot_busy / (1024.0 * tot_act / 8.0), 4) if tot_act else None
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: v.get("mfma_utilisation") for k, v in out["kernels"].items()}), out["gemm_family_mfma_utilisation"])
