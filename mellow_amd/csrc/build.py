#!/usr/bin/env python3
"""Build libmellow_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python mellow_amd/csrc/build.py [--force] [--verbose]

Objects are cached under mellow_amd/csrc/build/ (git-ignored) keyed on source mtimes; the shared
library lands in mellow_amd/lib/ (git-ignored, but shipped to the GPU box by gpurun)."""
from __future__ import annotations

import argparse
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
LIBDIR = os.path.join(PKG, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libmellow_hip.so")
SOURCES = ["gemm_f32.hip", "gemm_fp8.hip", "gemm_bf16x3.hip", "decode.hip", "prefill_attn.hip", "encoder.hip", "stft_fft.hip", "engine.cpp", "engine_weights.cpp", "engine_encoder.cpp", "engine_lm.cpp", "engine_dev.cpp"]
HEADERS = ["common.h", "kernels.h", "gemm_epilogue.h", "engine_internal.h", os.path.join("..", "..", "include", "mellow_hip.h")]
ARCH = "gfx950"
FLAGS = (["-DMELLOW_KDEBUG"] if os.environ.get("MELLOW_KDEBUG") else []) + os.environ.get("MELLOW_EXTRA_FLAGS", "").split() + ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-x", "hip"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


# Per-file flags.  VGPR-form MFMA: keep accumulators in architectural VGPRs (gfx950's register file is unified).  Without it
# the compiler parks the accumulators of the bf16 / fp8 kernels in AGPRs and copies all 64 of them in and out of VGPRs every
# loop iteration (v_accvgpr_read/write: as many VALU issue slots as the MFMAs themselves), and every `O *= alpha` of the
# flash attention is a read-modify-write through copies.
VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]
FILE_FLAGS = {name: VGPR_FORM for name in os.environ.get("MELLOW_VGPR_FORM_FILES", "gemm_bf16x3.hip gemm_fp8.hip prefill_attn.hip encoder.hip").split()}
# Kernel-argument preload (gfx950): the command processor writes the first dwords of a kernel's argument block into SGPRs while it
# sets the dispatch up, so the first address computations of a launch do not wait for a scalar load from the argument buffer.
# The decode kernels take their hot pointers as leading scalar arguments for this (decode.hip); measured on one kernel of the
# step (gate/up, 30 of 124 launches): 48.2-48.5 -> 47.65 ms of decode per 63 steps, 0.3 us per launch.  MELLOW_KERNARG_PRELOAD=0: off.
KERNARG_PRELOAD = [] if os.environ.get("MELLOW_KERNARG_PRELOAD", "14") == "0" else ["-mllvm", "-amdgpu-kernarg-preload-count=" + os.environ.get("MELLOW_KERNARG_PRELOAD", "14")]
FILE_FLAGS["decode.hip"] = FILE_FLAGS.get("decode.hip", []) + KERNARG_PRELOAD


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(HERE, h) for h in HEADERS] + [os.path.abspath(__file__)]
    jobs = []
    objs = []
    for src in SOURCES:
        sp = os.path.join(HERE, src)
        op = os.path.join(OBJDIR, src + ".o")
        objs.append(op)
        if force or _stale(op, [sp] + hdrs):
            jobs.append((sp, op))

    def cc(job):
        sp, op = job
        cmd = [hipcc] + FLAGS + FILE_FLAGS.get(os.path.basename(sp), []) + ["-c", sp, "-o", op]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        return sp, r

    with cf.ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        for sp, r in ex.map(cc, jobs):
            if r.returncode != 0:
                sys.stderr.write(r.stdout + r.stderr)
                raise RuntimeError(f"hipcc failed on {sp}")
            if verbose and r.stderr.strip():
                sys.stderr.write(r.stderr)
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link failed")
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    print(build(a.force, a.verbose))
