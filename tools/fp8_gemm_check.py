#!/usr/bin/env python3
"""Developer tool (GPU box): the fp8 GEMM against a torch emulation of the SAME quantisation (per-row e4m3 of A,
per-row e4m3 of W, exact products, fp32-ish accumulation), and its speed."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mellow_amd.engine import Engine  # noqa: E402


def quant_rows(x: torch.Tensor):
    amax = x.abs().amax(dim=1, keepdim=True)
    scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    inv = torch.where(amax > 0, 448.0 / amax, torch.zeros_like(amax))
    q = (x * inv).to(torch.float8_e4m3fn).to(torch.float32)
    return q, scale


def main():
    eng = Engine(device=0, max_positions=1024, precision="fp8")
    torch.manual_seed(0)
    for M, N, K in ((200, 132, 64), (389, 576, 576), (1000, 960, 1536), (12448, 3072, 576)):
        A = torch.randn(M, K) * torch.rand(M, 1) * 3
        W = torch.randn(N, K) * 0.05 * (1 + torch.rand(N, 1))
        Cc, ms = eng.debug_gemm_fp8(A, W, iters=10)
        Aq, sa = quant_rows(A)
        Wq, sw = quant_rows(W)
        ref = (Aq.double() @ Wq.double().T) * sa.double() * sw.double().T
        exact = A.double() @ W.double().T
        d = (Cc.double() - ref).abs().max().item()
        print(f"M {M} N {N} K {K}: max|gpu - emulation| {d:.3e} (rel {d / ref.abs().max().item():.2e}); "
              f"quantisation error vs exact fp32 GEMM rel {((ref - exact).abs().max() / exact.abs().max()).item():.3e}; "
              f"quant {ms[0] * 1e3:.1f} us, gemm {ms[1] * 1e3:.1f} us = {2.0 * M * N * K / ms[1] / 1e9:.1f} TFLOP/s "
              f"({2.0 * M * N * K / (ms[0] + ms[1]) / 1e9:.1f} with the quantisation pass)")


if __name__ == "__main__":
    main()
