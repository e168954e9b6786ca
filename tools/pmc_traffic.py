#!/usr/bin/env python3
"""Turn two rocprofv3 counter-collection CSVs (one --pmc FETCH_SIZE pass, one --pmc WRITE_SIZE pass of
tools/pmc_prefill.py) into the per-launch memory-side traffic of the gemm_f32_kernel family
(profiles/rNN_pmc_gemm_traffic.json).  FETCH_SIZE / WRITE_SIZE are in KiB; the gfx950 x2 correction of
/opt/skills/guides/MI355X_MICROARCH.md (HBM section: 128-B read requests tallied at 64 B) is applied to reads.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_sha16  # noqa: E402  (bench.py refuses to print a result measured on other kernel sources)


def family(path, counter):
    n, tot = 0, 0.0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and ("gemm_f32_kernel" in r["Kernel_Name"] or "gemm_x3p_kernel" in r["Kernel_Name"] or "gemm_x3q_kernel" in r["Kernel_Name"]
                                             or "gemm_bf16x3f_kernel" in r["Kernel_Name"] or "stft_fft_power_kernel" in r["Kernel_Name"]):
            n += 1
            tot += float(r["Counter_Value"])
    return n, tot


nf, fetch = family(sys.argv[1], "FETCH_SIZE")
nw, write = family(sys.argv[2], "WRITE_SIZE")
assert nf == nw and nf > 0, (nf, nw)
out = {
    "kernel": "dense GEMM family of encoder + LM prefill: gemm_x3q_kernel (LM prefill) / gemm_x3p_kernel / gemm_bf16x3f_kernel (f32x3 mode) + gemm_f32_kernel (all instances) + stft_fft_power_kernel (the STFT of the f32x3 mode)",
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, tools/pmc_prefill.py "
              "(2 encoder+prefill passes at B=32, in the `precision` mode below); reduced with tools/pmc_traffic.py",
    "source_sha16": kernel_source_sha16(),
    "precision": os.environ.get("MELLOW_PRECISION", "f32x3"),      # the mode tools/pmc_prefill.py ran in
    "launches": nf,
    "fetch_size_kb_per_launch": fetch / nf,
    "write_size_kb_per_launch": write / nw,
    "fetch_bytes_per_launch_x2_corrected": 2.0 * fetch * 1024.0 / nf,
    "write_bytes_per_launch": write * 1024.0 / nw,
    "traffic_bytes_per_launch": (2.0 * fetch + write) * 1024.0 / nf,
    "note": "gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md HBM section): doubled; "
            "memory-side requests include Infinity-Cache hits",
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
