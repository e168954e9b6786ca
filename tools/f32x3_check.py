#!/usr/bin/env python3
"""Developer tool (GPU box): accuracy (against an fp64 product) and speed of the three fp32 GEMM paths:
the exact fp32 MFMA kernel, and the bf16x3 split kernel with 9 and with 6 partial products."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mellow_amd.engine import Engine  # noqa: E402

eng = Engine(device=0, max_positions=1024)
torch.manual_seed(0)
for M, N, K in ((389, 576, 576), (1000, 960, 1536), (12448, 3072, 576), (12448, 576, 1536), (16384, 2048, 2048)):
    A = torch.randn(M, K) * (0.2 + 3 * torch.rand(M, 1))
    W = torch.randn(N, K) * 0.05 * (1 + torch.rand(N, 1))
    exact = A.double() @ W.double().T
    scale = exact.abs().max().item()
    cpu = (A @ W.T).double()
    line = f"M {M} N {N} K {K}: torch-CPU fp32 max err {(cpu - exact).abs().max().item() / scale:.2e} |"
    for mode, name in ((0, "fp32 MFMA"), (6, "bf16x3 6-term"), (16, "bf16x3 fused")):
        Cc, ms = eng.debug_gemm_f32(A, W, mode=mode, iters=10)
        err = (Cc.double() - exact).abs()
        tf = 2.0 * M * N * K / ms[1] / 1e9
        tf_all = 2.0 * M * N * K / (ms[0] + ms[1]) / 1e9
        line += f" {name}: max {err.max().item() / scale:.2e} rms {err.pow(2).mean().sqrt().item() / scale:.2e}, {ms[1] * 1e3:.0f} us = {tf:.0f} TF ({tf_all:.0f} incl. split) |"
    print(line, flush=True)
