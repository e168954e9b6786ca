#!/usr/bin/env python3
"""Turn two rocprofv3 counter-collection CSVs (one --pmc FETCH_SIZE pass, one --pmc WRITE_SIZE pass) into the memory-side traffic
of a kernel family (profiles/rNN_pmc_*_traffic.json).  FETCH_SIZE / WRITE_SIZE are in KiB; the gfx950 x2 correction of
/opt/skills/guides/MI355X_MICROARCH.md (HBM section: 128-B read requests tallied at 64 B) is applied to reads.

    python tools/pmc_traffic.py gemm   <fetch.csv> <write.csv> <out.json>     workload tools/pmc_prefill.py  -> bytes per GEMM launch
    python tools/pmc_traffic.py decode <fetch.csv> <write.csv> <out.json>     workload tools/pmc_decode.py -> bytes per decode step"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_sha16  # noqa: E402  (bench.py refuses to print a result measured on other kernel sources)

GEMM = ("gemm_f32_kernel", "gemm_x3p_kernel", "gemm_x3q_kernel", "gemm_x3w_kernel", "gemm_bf16x3f_kernel", "stft_fft_power_kernel",
        "gemm_mx8_kernel", "quant_mx8_kernel", "splitk_finish_kernel")
HELPERS = ("splitk_finish_kernel", "quant_mx8_kernel")     # the second launch of a split-K GEMM / the standalone quantiser in front of an
                                                           # fp8 GEMM: their bytes count, they are not launches of their own
DECODE = ("dec_qkv_kernel", "dec_qkv2_kernel", "dec_attn_kernel", "dec_oproj_kernel", "dec_gateup16_kernel", "dec_down_kernel",
          "dec_final_norm_kernel", "dec_fullk_kernel", "dec_head3_kernel", "dec_head3r_kernel", "dec_qkv2x3_kernel",
          "dec_gateup3_kernel", "dec_argmax_kernel", "dec_compact_kernel")


def family(path, counter, names):
    per = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = next((n for n in names if n in r["Kernel_Name"]), None)
        if k is None:
            continue
        a = per.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return per


mode, fpath, wpath, opath = sys.argv[1:5]
names = GEMM if mode == "gemm" else DECODE
pf, pw = family(fpath, "FETCH_SIZE", names), family(wpath, "WRITE_SIZE", names)
nf, fetch = sum(v[0] for k, v in pf.items() if k not in HELPERS), sum(v[1] for v in pf.values())
nw, write = sum(v[0] for k, v in pw.items() if k not in HELPERS), sum(v[1] for v in pw.values())
assert nf == nw and nf > 0, (nf, nw)
out = {
    "source_sha16": kernel_source_sha16(),
    "precision": os.environ.get("MELLOW_PRECISION", "f32x3"),      # the mode the workload ran in
    "launches": nf,
    "note": "gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads (MI355X_MICROARCH.md HBM section): doubled; "
            "memory-side requests include Infinity-Cache hits",
    "per_kernel_bytes_per_launch": {k: round((2.0 * pf[k][1] + pw.get(k, [0, 0.0])[1]) * 1024.0 / pf[k][0]) for k in pf},
}
if mode == "gemm":
    out.update({
        "kernel": "dense GEMM family of encoder + LM prefill: gemm_x3q_kernel (LM prefill, Swin stages 2-3) / gemm_x3w_kernel (stage 0 qkv, fc1) / "
                  "gemm_x3p_kernel / gemm_bf16x3f_kernel (f32x3 mode) + gemm_f32_kernel (all instances) + stft_fft_power_kernel (the STFT of "
                  "the f32x3 mode); fp8 mode: gemm_mx8_kernel (+ quant_mx8_kernel where no producer emits the AMX image); "
                  "splitk_finish_kernel's / quant_mx8_kernel's bytes are counted with the launch they serve",
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, tools/pmc_prefill.py (2 encoder+prefill passes at "
                  "B=32, in the `precision` mode); reduced with tools/pmc_traffic.py gemm",
        "fetch_bytes_per_launch_x2_corrected": 2.0 * fetch * 1024.0 / nf,
        "write_bytes_per_launch": write * 1024.0 / nw,
        "traffic_bytes_per_launch": (2.0 * fetch + write) * 1024.0 / nf,
    })
else:
    steps = pf["dec_argmax_kernel"][0]         # one arg-max launch per decode step (the prefill's last-position step included)
    out.update({
        "kernel": "every kernel of the decode step (dec_*), summed per step",
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, tools/pmc_decode.py (one call at B=32, max_len 6: "
                  "the prefill's last-position step + 5 decode steps, contexts 389..393); reduced with tools/pmc_traffic.py decode",
        "steps": steps,
        "fetch_bytes_per_step_x2_corrected": 2.0 * fetch * 1024.0 / steps,
        "write_bytes_per_step": write * 1024.0 / steps,
        "traffic_bytes_per_step": (2.0 * fetch + write) * 1024.0 / steps,
    })
json.dump(out, open(opath, "w"), indent=1)
print(json.dumps(out))
