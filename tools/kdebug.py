#!/usr/bin/env python3
"""Developer tool (GPU box): s_memtime phase stamps of workgroup 0 of the decode kernels."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mellow_amd import synth  # noqa: E402
from mellow_amd.engine import Engine  # noqa: E402

eng = Engine(device=0)
eng.load_state_dict(synth.make_state_dict(0))
B = 32
a1, a2, ids = synth.make_batch(B)
a1d, a2d, idsd = eng._f32(a1), eng._f32(a2), eng._i32(ids)
eng.generate(a1d, a2d, idsd, max_len=8, stop_id=0, ignore_stop=True)
fn = eng.lib.mellow_dev_kdebug
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
eng.set_graph(False)
assert fn(eng.h, 1, None) == 0
eng.generate(a1d, a2d, idsd, max_len=8, stop_id=0, ignore_stop=True)
out = (C.c_uint64 * 64)()
assert fn(eng.h, 0, out) == 0
v = np.asarray(list(out), dtype=np.int64).reshape(8, 8)
names = {0: "dec_qkv      [start, issued, mfma done, reduced(sync), pre-store]",
         1: "dec_attn     [start, issued, sync1(slab sums in LDS), sync2(rope/lds), kv loop done, sync3, end]",
         2: "dec_oproj    [start, issued, merge+mfma done, sync, end]",
         3: "dec_gateup   [start, issued, mfma done, sync, end]",
         4: "dec_down     [start, issued, mfma done, sync, end]",
         5: "dec_lm_head  [start, issued, mfma done, sync, -]"}
names[7] = "dec_qkv2     [start, issued, mfma done, sync, end]  (h-part workgroup)"
for k, nm in names.items():
    row = v[k]
    base = row[0]
    print(nm)
    print("   cycles since start:", [int(x - base) if x else None for x in row[:7]])

r = v[6]
print("qkv startup probe (cycles): entry->kernarg", int(r[1]-r[0]), " kernarg->issued", int(r[2]-r[1]), " issued->first data", int(r[3]-r[2]))

# absolute order of the last launches (same counter on every CU): gaps between one kernel's last stamp and the next one's first
ev = []
for k in (7, 1, 2, 3, 4, 5):
    st = [int(x) for x in v[k][:7] if x]
    if st:
        ev.append((min(st), max(st), names[k].split()[0]))
ev.sort()
t0 = ev[0][0]
print("timeline (cycles, workgroup 0 of each kernel):", [(n, a - t0, b - t0) for a, b, n in ev])

# ---- spans of every launch of ONE decode step (earliest workgroup start / latest workgroup end, 100 MHz clock) and the gaps between
#      consecutive launches: what a dependent boundary costs, measured from inside (library built with -DMELLOW_KDEBUG) ----
sp = eng.lib.mellow_dev_kdebug_spans
sp.restype = C.c_int
sp.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
pre = eng.prefix(a1d, a2d, idsd)
logits = eng.lm_prefill(pre, reserve=8)
tok = logits.argmax(-1).to("cpu").numpy().astype(np.int32)
for _ in range(3):
    logits = eng.lm_decode_step(tok)          # warm
assert fn(eng.h, 1, None) == 0
eng.lm_decode_step(tok)
assert fn(eng.h, 0, out) == 0
spans = (C.c_uint64 * 960)()
assert sp(eng.h, spans) == 0
s = np.asarray(list(spans), dtype=np.uint64).reshape(480, 2)
kinds = ["qkv|qkv2(prev)", "attn", "o_proj", "gate/up", "qkv2|down"]
rows = []
for i in range(480):
    if s[i, 1] == 0 or s[i, 0] == np.uint64(0xFFFFFFFFFFFFFFFF):
        continue
    name = kinds[i % 5] if i < 150 else ["final_norm", "lm_head", "argmax"][i - 150] if i < 153 else "?"
    rows.append((i, name, int(s[i, 0]), int(s[i, 1])))
t0 = rows[0][2]
print("launch spans of one decode step (us; gap = this launch's first workgroup start - previous launch's last workgroup end):")
tot_gap = tot_span = 0.0
per = {}
prev_end = None
for i, name, a, b in rows:
    span = (b - a) * 0.01
    gap = (a - prev_end) * 0.01 if prev_end is not None else 0.0
    prev_end = b
    tot_gap += gap; tot_span += span
    d = per.setdefault(name, [0, 0.0, 0.0]); d[0] += 1; d[1] += span; d[2] += gap
    if i < 12 or i >= 145:
        print(f"  #{i:3d} {name:16s} start {(a - t0) * 0.01:8.2f} end {(b - t0) * 0.01:8.2f} span {span:6.2f} gap {gap:6.2f}")
print(f"step: {len(rows)} launches, {(rows[-1][3] - t0) * 0.01:.1f} us from first start to last end; sum of spans {tot_span:.1f}, sum of gaps {tot_gap:.1f}")
for name, (n, sp_, gp) in per.items():
    print(f"  {name:16s} n={n:3d} mean span {sp_ / n:6.2f} us, mean gap before it {gp / n:6.2f} us")
