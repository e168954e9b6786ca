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


def quant_mx8(x: torch.Tensor):
    """the engine's activation format (mellow_amd/csrc/common.h: amx_store_block), dequantised"""
    M, K = x.shape
    Kp = (K + 31) // 32 * 32
    xb = torch.zeros(M, Kp)
    xb[:, :K] = x
    xb = xb.view(M, Kp // 32, 32)
    amax = xb.abs().amax(-1, keepdim=True)
    mant, ex = torch.frexp(amax / 448.0)
    e = torch.where(mant > 0.5, ex, ex - 1)
    scale = torch.where(amax > 0, torch.ldexp(torch.ones_like(amax), e), torch.ones_like(amax))
    return ((xb / scale).to(torch.float8_e4m3fn).to(torch.float32) * scale).view(M, Kp)[:, :K]


def main():
    eng = Engine(device=0, max_positions=1024, precision="fp8")
    torch.manual_seed(0)
    for M, N, K in ((200, 132, 64), (389, 576, 576), (1000, 960, 1536), (12448, 3072, 576)):
        A = torch.randn(M, K) * torch.rand(M, 1) * 3
        W = torch.randn(N, K) * 0.05 * (1 + torch.rand(N, 1))
        Cc, ms = eng.debug_gemm_fp8(A, W, iters=10)
        Aq = quant_mx8(A)                       # MXFP8 activations (one power-of-two scale per 32 k), dequantised
        Wq, sw = quant_rows(W)
        ref = (Aq.double() @ Wq.double().T) * sw.double().T
        exact = A.double() @ W.double().T
        d = (Cc.double() - ref).abs().max().item()
        print(f"M {M} N {N} K {K}: max|gpu - emulation| {d:.3e} (rel {d / ref.abs().max().item():.2e}); "
              f"quantisation error vs exact fp32 GEMM rel {((ref - exact).abs().max() / exact.abs().max()).item():.3e}; "
              f"quant {ms[0] * 1e3:.1f} us, gemm {ms[1] * 1e3:.1f} us = {2.0 * M * N * K / ms[1] / 1e9:.1f} TFLOP/s "
              f"({2.0 * M * N * K / (ms[0] + ms[1]) / 1e9:.1f} with the quantisation pass)")


if __name__ == "__main__":
    main()
