#!/usr/bin/env python3
"""Developer tool (GPU box): time of the f32x3 (mode 16) and exact fp32 (mode 0) GEMM kernels on the LM prefill shapes
and a few encoder shapes; random data.  `python tools/f32x3_bench.py [MxNxK ...] [--modes 0,16] [--iters 20]`
Also the workload for `rocprofv3 --pmc ...` passes on the GEMM alone."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mellow_amd.engine import Engine  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
opts = dict(a[2:].split("=") for a in sys.argv[1:] if a.startswith("--") and "=" in a)
modes = [int(m) for m in opts.get("modes", "0,16").split(",")]
iters = int(opts.get("iters", "20"))
shapes = [tuple(int(v) for v in a.split("x")) for a in args] or [
    (12448, 3072, 576), (12448, 960, 576), (12448, 576, 576), (12448, 576, 1536), (16384, 1536, 384), (65536, 768, 192)]
eng = Engine(device=0)
torch.manual_seed(0)
for M, N, K in shapes:
    A = torch.randn(M, K) * (0.2 + 3 * torch.rand(M, 1))
    W = torch.randn(N, K) * 0.05 * (1 + torch.rand(N, 1))
    line = f"M {M:6d} N {N:5d} K {K:5d}:"
    for mode in modes:
        _, ms = eng.debug_gemm_f32(A, W, mode=mode, iters=iters)
        line += f"  mode {mode:2d}: {ms[1] * 1e3:7.1f} us {2.0 * M * N * K / ms[1] / 1e9:6.1f} TF"
    print(line, flush=True)
