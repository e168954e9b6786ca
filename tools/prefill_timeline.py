#!/usr/bin/env python3
"""Developer tool (CPU): timeline of the LAST encoder + prefill phase in a `rocprofv3 --kernel-trace` CSV of tools/decode_probe.py:
per kernel family the summed duration, the time it ran alone / beside another kernel, and the phase spans.
`python tools/prefill_timeline.py trace.csv [--dump]`"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r["Grid_Size_X"]) for r in rows]
# passes start with reflect_pad_kernel
starts = [i for i, k in enumerate(ks) if "reflect_pad" in k[2]]
i0 = starts[-1]
i1 = next(i for i in range(i0, len(ks)) if "dec_" in ks[i][2] and "final" not in ks[i][2])
seg = ks[i0:i1]
t0 = seg[0][0]
ipre = next(i for i, k in enumerate(seg) if "prefix_assemble" in k[2])
print(f"encoder: {len(seg[:ipre])} launches, {(seg[ipre][0] - t0) / 1e3:.1f} us;  prefill: {len(seg[ipre:])} launches, {(max(k[1] for k in seg) - seg[ipre][0]) / 1e3:.1f} us")


def fam(n):
    for key in ("gemm_x3q_kernel<0>", "gemm_x3q_kernel<3>", "gemm_x3q_kernel<4>", "gemm_x3r", "gemm_x3p", "gemm_x3w", "prefill_attention", "window_attention", "layernorm", "rmsnorm",
                "splitk_finish", "stft", "fold_patch", "tail_", "gemm_bf16x3f"):
        if key in n:
            return key
    return n.split("(")[0][-40:]


for name, part in (("encoder", seg[:ipre]), ("prefill", seg[ipre:])):
    ev = []
    for s, e, n, q, g in part:
        ev.append((s, 1, n)); ev.append((e, -1, n))
    ev.sort()
    tot = {}
    for s, e, n, q, g in part:
        f = fam(n)
        d = tot.setdefault(f, [0, 0.0, 0.0])
        d[0] += 1; d[1] += (e - s) / 1e3
    # alone time per family: sweep
    active = []
    last = ev[0][0]
    idle = 0.0
    for t, d, n in ev:
        dt = (t - last) / 1e3
        if dt > 0:
            if len(active) == 1:
                tot[fam(active[0])][2] += dt
            elif not active:
                idle += dt
        last = t
        if d == 1:
            active.append(n)
        else:
            active.remove(n)
    span = (max(k[1] for k in part) - part[0][0]) / 1e3
    print(f"-- {name}: span {span:.1f} us, idle {idle:.1f} us")
    for f, (c, d, a) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f"   {f:40s} n={c:4d} sum {d:9.1f} us  avg {d / c:7.1f}  alone {a:9.1f} us")
if "--dump" in sys.argv:
    for s, e, n, q, g in seg[ipre:ipre + 40]:
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} q{q} grid {g:>8s} {fam(n)}")
