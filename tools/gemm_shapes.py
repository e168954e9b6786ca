#!/usr/bin/env python3
"""Developer tool (GPU box): per-shape time / TFLOP/s of every fp32 MFMA GEMM launch of one generate() call
(HIP events of the engine profiler; serialised launches, so times include launch gaps)."""
import collections
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _opts import engine_options  # noqa: E402  (--opt KEY=VALUE -> engine options)
OPTS = engine_options()
from mellow_amd import synth  # noqa: E402
from mellow_amd.engine import Engine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
PREC = sys.argv[2] if len(sys.argv) > 2 else "f32"
eng = Engine(device=0, precision=PREC, options=OPTS)
eng.load_state_dict(synth.make_state_dict(0))
a1, a2, ids = synth.make_batch(B)
a1d, a2d, idsd = eng._f32(a1), eng._f32(a2), eng._i32(ids)
for _ in range(2):
    eng.generate(a1d, a2d, idsd, max_len=2, stop_id=0, ignore_stop=True)
eng.prof_enable(True)
eng.prof_reset()
eng.generate(a1d, a2d, idsd, max_len=2, stop_id=0, ignore_stop=True)
out = os.path.join(ROOT, "gpurun_out", "prof_dump.csv")
os.makedirs(os.path.dirname(out), exist_ok=True)
fn = eng.lib.mellow_dev_prof_dump
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_char_p]
assert fn(eng.h, out.encode()) == 0
agg = collections.OrderedDict()
for line in open(out).read().split("\n")[1:]:
    if not line:
        continue
    fam, M, N, K, epi, ms, fl = line.split(",")
    if int(fam) != 0:
        continue
    k = (int(M), int(N), int(K), int(epi))
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1; a[1] += float(ms); a[2] += float(fl)
tot = sum(a[1] for a in agg.values())
print(f"{'M':>8s} {'N':>5s} {'K':>5s} epi  n   total_ms  avg_us   TF/s   tiles(128x128)  share")
for (M, N, K, epi), (n, ms, fl) in agg.items():
    tiles = -(-M // 128) * -(-N // 128)
    print(f"{M:8d} {N:5d} {K:5d} {epi:3d} {n:3d} {ms:9.3f} {ms / n * 1e3:8.1f} {fl / ms / 1e9:6.1f}   {tiles:6d} ({tiles / 512:5.2f} rounds) {ms / tot * 100:5.1f}%")
print("total GEMM ms", round(tot, 3))
