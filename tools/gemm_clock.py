#!/usr/bin/env python3
"""Developer tool (GPU box, library built with MELLOW_KDEBUG=1): inside the fp32 MFMA GEMM main loop of a mid-grid
workgroup -- effective shader clock (s_memtime vs the 100 MHz s_memrealtime), clocks per k-tile, and how they split
between the compute section and the barrier / prefetch-issue section."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mellow_amd.engine import Engine  # noqa: E402

eng = Engine(device=0, max_positions=1024)
kd = eng.lib.mellow_dev_kdebug
kd.restype = C.c_int
kd.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
gt = eng.lib.mellow_dev_gemm_time
gt.restype = C.c_int
gt.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(16384, 2048, 576), (16384, 2048, 2048), (12448, 3072, 576)]
for M, N, K in shapes:
    ms = C.c_float(0)
    assert kd(eng.h, 1, None) == 0
    assert gt(eng.h, M, N, K, 3, C.byref(ms)) == 0
    out = (C.c_uint64 * 64)()
    assert kd(eng.h, 0, out) == 0
    v = np.asarray(list(out), dtype=np.int64)
    cyc, ticks, kt, grid, tc, ts, hwid = v[56:63]
    mhz = cyc / max(ticks, 1) * 100.0
    print(f"{M}x{N}x{K}: {2.0 * M * N * K / ms.value / 1e9:6.1f} TFLOP/s (instrumented) | main loop {cyc} clocks @ {mhz:.0f} MHz, "
          f"{cyc / max(kt, 1):.0f} per k-tile (MFMA-bound: 8192 with 2 workgroups per CU) = compute+store {tc / max(kt, 1):.0f} "
          f"+ barrier/prefetch-issue {ts / max(kt, 1):.0f} | wave slot {hwid & 15} simd {(hwid >> 4) & 3}")
