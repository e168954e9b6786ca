#!/usr/bin/env python3
"""Developer tool (GPU box): TFLOP/s of the fp32 MFMA GEMM on chosen (M, N, K) shapes (synthetic buffers)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mellow_amd import synth  # noqa: E402
from mellow_amd.engine import Engine  # noqa: E402

eng = Engine(device=0, max_positions=1024)
fn = eng.lib.mellow_dev_gemm_time
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [
    (16384, 2048, 576), (65536, 2048, 576), (16384, 2048, 2048), (65536, 1024, 1024),
    (12448, 3072, 576), (12448, 960, 576), (12448, 576, 576), (12448, 576, 1536), (12544, 3072, 576)]
for M, N, K in shapes:
    ms = C.c_float(0)
    assert fn(eng.h, M, N, K, 10, C.byref(ms)) == 0, eng.last_error() if hasattr(eng, "last_error") else "error"
    tiles = -(-M // 128) * -(-N // 128)
    print(f"M {M:7d} N {N:5d} K {K:5d}: {ms.value * 1e3:9.1f} us  {2.0 * M * N * K / ms.value / 1e9:7.1f} TFLOP/s   tiles {tiles} ({tiles / 512:.2f} rounds)")
