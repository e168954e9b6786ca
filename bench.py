#!/usr/bin/env python3
"""Headline benchmark: audio-pair responses/sec (+ p50 first-token latency) of the Mellow hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): v0 167M, batch 32 per GPU,
2 x 10 s / 32 kHz synthetic clips + 16-token prompt per example, max_len = 64, greedy decode, fixed-length mode
(stop id ignored so every step runs: deterministic work).  One "step" = one pass of the whole hot path
(log-mel front-end -> HTSAT encoder -> projection -> prefix -> KV-cached prefill + 63 decode steps) over one batch,
inputs already resident in HBM.  Weights: seeded synthetic checkpoint with the real state_dict layout
(the real v0.ckpt cannot be fetched offline).  N > 1: weak scaling, one process per GPU, every rank runs its own
32 examples and the token ids are all-gathered once per step over RCCL.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline     the dominant phase by time, the decode step (HBM-bound): algorithmic bytes / step time, + PMC traffic per step
  roofline_gemm  the dominant family by FLOPs (dense GEMMs of encoder + prefill, MFMA-bound), HIP events on the engine's stream
  cpu_baseline the oracle (op-for-op CPU port of the reference: no KV cache) timed on a bounded sample (N = 1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before the first GPU call: see mellow_amd/__init__.py

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_FP8_MFMA_TFLOPS = 5000.0     # dense fp8 matrix peak (same guide): the block-scaled K = 64 form gemm_mx8_kernel issues (4.6 PF measured there)
PEAK_HBM_GBS = 8000.0
PEAK_F32X3_TFLOPS = 2500.0 / 6.0   # f32x3 mode: six bf16 MFMA products per fp32 product on the 2.5 PF dense bf16 pipe
PMC_TRAFFIC_FILE = "r06_pmc_gemm_traffic.json"          # dense GEMM family (tools/pmc_traffic.py gemm)
PMC_DECODE_FILE = "r06_pmc_decode_traffic.json"         # decode step (tools/pmc_traffic.py decode)
FFT_GFLOP_PER_CLIP = 0.051         # algorithmic cost of the STFT; the kernel runs it as a dense DFT GEMM (2.10 GF/clip)

# algorithmic work per response at max_len = 64, prefix 389
ENCODER_GFLOP_PER_RESPONSE = 24.33                  # 2 clips of 10 s: one 1024-frame crop each
PREFILL_GFLOP_PER_RESPONSE = 87.90
DENSE_GFLOP_PER_RESPONSE = ENCODER_GFLOP_PER_RESPONSE + PREFILL_GFLOP_PER_RESPONSE          # encoder + LM prefill -> MFMA-bound part


def dense_gflop_per_response(clip_seconds: int) -> float:
    """clips longer than 10 s run the encoder's long path (reference htsat.py:908-936): 7 crops of 1024 frames per
    30 s clip, each a full encoder pass, averaged afterwards -- 7x the encoder FLOPs; the LM prefill (389 positions) is unchanged."""
    crops = 1 if clip_seconds <= 10 else 7
    return crops * ENCODER_GFLOP_PER_RESPONSE + PREFILL_GFLOP_PER_RESPONSE


_T0 = time.time()


def _progress(msg: str):
    """leg-by-leg progress on stderr (stdout carries exactly one JSON line)"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"# [{time.time() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def kernel_source_sha16() -> str:
    """sha256 over the kernel + engine sources: committed PMC results carry it, so a stale file is never printed."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "mellow_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def committed_pmc(fname: str, precision: str, key: str):
    """HBM-side traffic from a committed rocprofv3 PMC result (FETCH_SIZE and WRITE_SIZE in separate passes, gfx950 x2 read
    correction applied: /opt/skills/guides/MI355X_MICROARCH.md): it cannot be sampled from inside this process.  The file
    carries a hash of the kernel sources and the numeric mode; a result measured on other sources is never printed."""
    try:
        with open(os.path.join(ROOT, "profiles", fname)) as f:
            pj = json.load(f)
    except Exception:
        return None, f"no PMC result committed (profiles/{fname})"
    if pj.get("precision", "f32x3") != precision:
        return None, f"profiles/{fname} was measured in the {pj.get('precision', 'f32x3')} mode, this run is {precision}: not printed"
    if pj.get("source_sha16") != kernel_source_sha16():
        return None, (f"profiles/{fname} was measured on other kernel sources (sha {pj.get('source_sha16')} != "
                      f"{kernel_source_sha16()}): not printed; regenerate with tools/collect_profiles.sh")
    return round(pj[key]), f"bytes (memory-side, PMC, profiles/{fname}, same kernel sources)"


def decode_algorithmic_bytes(B: int, L: int, T0: int = 389) -> float:
    """SURVEY.md 8(d), fp32 exact mode: per decode step the 538.06 MB of weights once, plus per example 46,080 B for every
    cached token read and for the one appended; steps 1..L-1 (token 0 comes from the prefill)."""
    w = 134_515_008 * 4.0
    return sum(w + B * 46080.0 * (T0 + i + 1) for i in range(1, L))


def decode_algorithmic_bytes_fp8(B: int, L: int, T0: int = 389) -> float:
    """The fp8 mode's own byte count: e4m3 weights (134.5 MB per step) and bf16 K/V shadow pages (23,040 B per cached token per
    example, read for every cached token and written once) -- what ITS decode step has to move, not the fp32 mode's."""
    w = 134_515_008 * 1.0
    return sum(w + B * 23040.0 * (T0 + i + 1) for i in range(1, L))


def cpu_baseline(max_len: int):
    """The oracle (CPU port of the reference algorithm, no KV cache) on a bounded sample, on ALL host cores:
    the thread count is chosen by timing a short run with min(32, cores) and with every core; B = 1 (encoder + prefix once,
    then 24 full re-forward decode steps) and B = 4 (12 steps); the remaining steps are extrapolated linearly in sequence
    length (the reference's per-step cost is proportional to 389+i)."""
    from mellow_amd import synth
    from oracle import mellow_oracle as O
    cores = os.cpu_count() or 1
    sd = synth.make_state_dict(0)
    lm = O.LMParams()
    a1, a2, ids = synth.make_batch(4)

    def run(B, n_meas):
        with torch.no_grad():
            t0 = time.time()
            prefix = O.generate_prefix_inference(sd, torch.from_numpy(a1[:B]), torch.from_numpy(a2[:B]), torch.from_numpy(ids[:B]))
            t_enc = time.time() - t0
            t0 = time.time()
            O.generate_batch(sd, lm, prefix, n_meas, 0.8, 1.0, -1, last_only=False)
            t_steps = time.time() - t0
        T0 = prefix.shape[1]
        per = t_steps / sum(T0 + i for i in range(n_meas))              # seconds per (step x sequence position) of the batch
        total = t_enc + per * sum(T0 + i for i in range(max_len))
        return {"responses_per_s": round(B / total, 5), "first_token_s": round(t_enc + per * T0, 3),
                "encode_s": round(t_enc, 3), "steps_measured": n_meas, "steps_s": round(t_steps, 3)}

    # Thread count: min(32, usable) is the working point (measured in round 1); every hardware thread is ALSO tried, in a
    # subprocess with a hard time limit, because on this pool's boxes os.cpu_count() reports 256 hardware threads while a
    # 256-thread torch run of the same 3-step sample does not finish in minutes (container CPU quota / oversubscription).
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else cores
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else round(int(q) / int(per), 2)
    except Exception:
        pass
    base = max(1, min(32, usable, int(quota) if quota else usable))
    torch.set_num_threads(base)
    t0 = time.time()
    run(1, 3)
    tried = {str(base): round(time.time() - t0, 3)}
    _progress(f"cpu_baseline: 3-step trial with {base} threads: {tried[str(base)]} s")
    threads = base
    if quota is not None and quota <= base:
        # the container is capped at `quota` CPUs: more threads than that only time-slice (measured once on this pool: the
        # 256-thread trial did not finish in 55 s against 0.5 s with 16 threads -- profiles/r02_bench.json of commit 5d0c0ce..)
        tried[str(usable)] = f"not tried: cgroup cpu.max caps the container at {quota} CPUs"
    elif usable > base:
        import subprocess
        code = ("import sys, time, torch; sys.path.insert(0, %r); import bench; from mellow_amd import synth; "
                "from oracle import mellow_oracle as O; torch.set_num_threads(%d); sd = synth.make_state_dict(0); "
                "a1, a2, ids = synth.make_batch(1); t0 = time.time()\n"
                "with torch.no_grad():\n"
                "    p = O.generate_prefix_inference(sd, torch.from_numpy(a1), torch.from_numpy(a2), torch.from_numpy(ids))\n"
                "    O.generate_batch(sd, O.LMParams(), p, 3, 0.8, 1.0, -1, last_only=False)\n"
                "print('TRIAL', time.time() - t0)") % (ROOT, usable)
        limit = 20.0 + 10.0 * tried[str(base)]
        try:
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=limit + 30.0)
            t_all = float(r.stdout.split("TRIAL")[1]) if "TRIAL" in r.stdout else None
        except subprocess.TimeoutExpired:
            t_all = None
        tried[str(usable)] = round(t_all, 3) if t_all is not None else f"did not finish within {limit + 30.0:.0f} s (incl. ~10 s set-up)"
        _progress(f"cpu_baseline: 3-step trial with {usable} threads: {tried[str(usable)]}")
        if t_all is not None and t_all < tried[str(base)]:
            threads = usable
    torch.set_num_threads(threads)
    b1 = run(1, 24)
    _progress(f"cpu_baseline: B=1 {b1}")
    b4 = run(4, 12)
    _progress(f"cpu_baseline: B=4 {b4}")
    best = max(b1["responses_per_s"], b4["responses_per_s"])
    return {
        "value": best, "unit": "responses/s", "cores": threads, "cores_available": cores, "cores_usable": usable,
        "cgroup_cpu_quota": quota, "threads_used": threads, "threads_tried_s": tried, "kind": "port",
        "sample": f"oracle (no KV cache, fp32) on {threads} of {cores} host threads (the faster of the counts tried on a 3-step run): "
                  f"B=1 encoder+prefix + 24 decode steps, B=4 encoder+prefix + 12 decode steps, each extrapolated linearly in "
                  f"sequence length to max_len={max_len}; value = the better of the two batch sizes",
        "b1": b1, "b4": b4, "first_token_s": b1["first_token_s"],
    }


def north_star_b64(L: int, precision: str):
    """The north_star's stated batch (64 examples per GPU, 2x10 s clips, max_len 64) on this GPU: 2 timed passes."""
    from mellow_amd import synth
    from mellow_amd.engine import Engine
    e = Engine(device=0, precision=precision)
    e.load_state_dict(synth.make_state_dict(0))
    a1, a2, ids = synth.make_batch(64)
    a1d, a2d, idsd = e._f32(a1), e._f32(a2), e._i32(ids)
    e.generate(a1d, a2d, idsd, max_len=L, stop_id=0, ignore_stop=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        _, _, _, ftm = e.generate(a1d, a2d, idsd, max_len=L, stop_id=0, ignore_stop=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    ph = e.last_phase_ms()
    e.close()
    peak = PEAK_F32X3_TFLOPS if precision == "f32x3" else PEAK_F32_MFMA_TFLOPS
    t_roof = 64 * (DENSE_GFLOP_PER_RESPONSE / (peak * 1e3)) + decode_algorithmic_bytes(64, L) / (PEAK_HBM_GBS * 1e9)
    return {"batch": 64, "value": round(64 / dt, 2), "unit": "responses/s", "ms_per_pass": round(dt * 1e3, 2),
            "first_token_ms": round(ftm, 2), "phase_ms": {k: round(v, 2) for k, v in ph.items()},
            "decode_ms_per_step": round(ph["decode_ms"] / (L - 1), 4),
            "decode_roofline": {"bound": "hbm", "achieved": round(decode_algorithmic_bytes(64, L) / (ph["decode_ms"] * 1e-3) / 1e9, 1),
                                "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                "frac": round(decode_algorithmic_bytes(64, L) / (ph["decode_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                "note": "two 32-row blocks: the f32x3 forms of the layer GEMM launches (weights loaded and split once for "
                                        "both blocks, activations pre-split by their producers) + the streaming lm_head (DESIGN 6e)"},
            "path_roofline_frac": round(t_roof * 1e3 / (dt * 1e3), 4)}


def configs3_leg(precision: str, B: int = 64, L: int = 128, clip_seconds: int = 30):
    """BASELINE configs[3] on this GPU (v0_s checkpoint layout = v0's, batch 64, 2 x 30 s clips -> the 7-crop long-mel encoder
    path, max_len 128): 2 timed passes, phase split, two-phase path roofline with the 7x encoder FLOPs of SURVEY 8d."""
    from mellow_amd import spec, synth
    from mellow_amd.engine import Engine
    e = Engine(device=0, precision=precision)
    e.load_state_dict(synth.make_state_dict(0))
    a1, a2, ids = synth.make_batch(B, n_samples=clip_seconds * spec.SAMPLE_RATE)
    a1d, a2d, idsd = e._f32(a1), e._f32(a2), e._i32(ids)
    e.generate(a1d, a2d, idsd, max_len=L, stop_id=0, ignore_stop=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ft = []
    for _ in range(2):
        _, _, _, ftm = e.generate(a1d, a2d, idsd, max_len=L, stop_id=0, ignore_stop=True)
        ft.append(ftm)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    ph = e.last_phase_ms()
    e.close()
    fp8 = precision == "fp8"
    peak = PEAK_FP8_MFMA_TFLOPS if fp8 else PEAK_F32X3_TFLOPS if precision == "f32x3" else PEAK_F32_MFMA_TFLOPS
    gf = dense_gflop_per_response(clip_seconds)
    dec_bytes = decode_algorithmic_bytes(B, L)
    t_mfma, t_hbm = B * gf / (peak * 1e3), dec_bytes / (PEAK_HBM_GBS * 1e9)
    enc_tf = B * 7 * ENCODER_GFLOP_PER_RESPONSE / ph["encode_ms"] if ph["encode_ms"] > 0 else 0.0      # GF / ms = TF/s
    return {"workload": f"v0_s layout (= v0), batch {B}, 2x{clip_seconds}s 32kHz synthetic clips (7 encoder crops per clip = {B * 2 * 7} crops), "
                        f"max_len={L}, greedy, fixed-length, 1 GPU",
            "value": round(B / dt, 2), "unit": "responses/s", "ms_per_pass": round(dt * 1e3, 2),
            "first_token_ms_p50": round(statistics.median(ft), 2), "phase_ms": {k: round(v, 2) for k, v in ph.items()},
            "decode_ms_per_step": round(ph["decode_ms"] / (L - 1), 4),
            "encoder_phase": {"achieved_tflops": round(enc_tf, 1), "peak": round(peak, 1), "frac": round(enc_tf / peak, 4),
                              "note": "7 x 24.33 GF per response (SURVEY 8d) / the front-end + encoder + prefix phase time"},
            "decode_roofline": {"bound": "hbm", "achieved": round(dec_bytes / (ph["decode_ms"] * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                "frac": round(dec_bytes / (ph["decode_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)},
            "path_roofline": {"t_roof_ms_per_pass": round((t_mfma + t_hbm) * 1e3, 3), "frac": round((t_mfma + t_hbm) / dt, 4),
                              "dense_gflop_per_response": round(gf, 2),
                              "definition": "F_dense/P_mfma(dtype) + Bytes_decode/8TB/s, F_dense = 7 x encoder + prefill (SURVEY 8d)"}}


ALT_NOTES = {
    "f32": "exact fp32 MFMA GEMMs (v_mfma_f32_32x32x2_f32), the mode the parity suite calls 'f32'; tokens identical to f32x3",
    "f32x3": "fp32 GEMMs as exact 3-way bf16 operand splits on the bf16 MFMA pipe; tokens identical to f32 (DESIGN 6c)",
    "fp8": "BASELINE config 5 numerics: e4m3 GEMMs in encoder + LM prefill, e4m3 weights and activations on the fp8 matrix pipe in the decode GEMM kernels; not bit-exact (DESIGN 6b)",
}


def alt_modes(B: int, L: int, headline: str):
    """Supplementary, never the headline: the other numeric modes on the same workload (3 passes each)."""
    from mellow_amd import synth
    from mellow_amd.engine import Engine
    sd = synth.make_state_dict(0)
    a1, a2, ids = synth.make_batch(B)
    res = {}
    for prec in ("f32", "f32x3", "fp8"):
        if prec == headline:
            continue
        e = Engine(device=0, precision=prec)
        e.load_state_dict(sd)
        a1d, a2d, idsd = e._f32(a1), e._f32(a2), e._i32(ids)
        e.generate(a1d, a2d, idsd, max_len=L, stop_id=0, ignore_stop=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            _, _, _, ftm = e.generate(a1d, a2d, idsd, max_len=L, stop_id=0, ignore_stop=True)
        torch.cuda.synchronize()
        res[prec] = {"value": round(3 * B / (time.perf_counter() - t0), 2), "unit": "responses/s",
                     "first_token_ms": round(ftm, 2), "phase_ms": {k: round(v, 2) for k, v in e.last_phase_ms().items()},
                     "note": ALT_NOTES[prec]}
        if prec == "fp8":
            # BASELINE configs[4]'s per-GPU batch (fp8 weights, B = 128, max_len 64): its single-GPU leg, 2 passes
            b1, b2, bi = synth.make_batch(128)
            b1d, b2d, bid = e._f32(b1), e._f32(b2), e._i32(bi)
            e.generate(b1d, b2d, bid, max_len=L, stop_id=0, ignore_stop=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                e.generate(b1d, b2d, bid, max_len=L, stop_id=0, ignore_stop=True)
            torch.cuda.synchronize()
            ph8 = e.last_phase_ms()
            kv16 = bool(e.describe()["options"]["fp8_kv16"]["value"])
            b8 = decode_algorithmic_bytes_fp8(128, L) if kv16 else (decode_algorithmic_bytes(128, L) - (L - 1) * 134_515_008 * 3.0)
            res["fp8_b128"] = {"batch": 128, "value": round(2 * 128 / (time.perf_counter() - t0), 2), "unit": "responses/s",
                               "phase_ms": {k: round(v, 2) for k, v in ph8.items()},
                               "decode_ms_per_step": round(ph8["decode_ms"] / (L - 1), 4),
                               "decode_roofline": {"bound": "hbm", "achieved": round(b8 / (ph8["decode_ms"] * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS,
                                                   "unit": "GB/s", "frac": round(b8 / (ph8["decode_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                                   "bytes_per_step": round(b8 / (L - 1)),
                                                   "note": "algorithmic bytes of THIS mode: 134.5 MB of e4m3 weights per step + "
                                                           + ("23,040 B (bf16 shadow pages)" if kv16 else "46,080 B (fp32 pages)")
                                                           + " per cached token per example"},
                               "note": "BASELINE configs[4] (fp8, B = 128, max_len 64) on one GPU; e4m3 GEMMs in encoder + prefill and in the decode GEMM kernels (DESIGN 6b)"}
        e.close()
    return res


def pipelined(n_ctx: int, B: int, L: int, n_batches: int, precision: str = "f32x3"):
    """Supplementary: n_ctx engine contexts (ONE weight copy) on one GPU, n_batches batches of B dealt round-robin (mellow_amd/serve.py)."""
    from mellow_amd import synth
    from mellow_amd.serve import EnginePool
    pool = EnginePool(synth.make_state_dict(0), n_contexts=n_ctx, device=0, precision=precision)
    batches = []
    for i in range(n_batches):
        a1, a2, ids = synth.make_batch(B, first=i * B)
        e = pool.engines[i % n_ctx]
        batches.append((e._f32(a1), e._f32(a2), e._i32(ids)))
    kw = dict(max_len=L, top_p=0.8, temperature=1.0, stop_id=0, ignore_stop=True)
    pool.generate_many(batches[: n_ctx], **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pool.generate_many(batches, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pool.close()
    return {"contexts": n_ctx, "batches": n_batches, "value": round(n_batches * B / dt, 2), "unit": "responses/s",
            "note": "throughput mode for a serving front-end: independent batches overlap on one GPU; per-batch latency is worse"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="examples per GPU")
    ap.add_argument("--max-len", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=("f32", "fp8", "f32x3"), default="f32x3",
                    help="f32x3 (default: fp32-accurate GEMMs as exact 3-way bf16 splits on the bf16 MFMA pipe; passes the whole "
                         "parity suite with the f32 tolerances and exact tokens), f32 (exact fp32 MFMA) or fp8 (BASELINE config 5: "
                         "e4m3 GEMMs in the encoder's Swin linears and LM prefill; a different metric line)")
    ap.add_argument("--no-alt-modes", action="store_true", help="skip the supplementary fp8 / f32x3 measurements")
    ap.add_argument("--no-b64", action="store_true", help="skip the supplementary batch-64 (north_star) measurement")
    ap.add_argument("--preset", choices=("configs1", "configs2", "configs3", "configs4"), default=None,
                    help="BASELINE.json configs[i] per-GPU shapes: configs1 = the headline (batch 32, max_len 64; the default), "
                         "configs2 = batch 32/GPU (256 over 8 GPUs), max_len 300, configs3 = batch 64, 2x30 s clips (the 7-crop long-mel "
                         "encoder path), max_len 128, configs4 = fp8 weights, batch 128/GPU, max_len 64")
    ap.add_argument("--clip-seconds", type=int, default=10, help="length of the synthetic clips (10, or 30 for the long-mel path)")
    ap.add_argument("--no-configs3", action="store_true", help="skip the supplementary BASELINE configs[3] measurement")
    ap.add_argument("--inflight", type=int, default=4,
                    help="also measure N engine contexts (one weight copy, mellow_engine_fork) pipelining independent batches on this "
                         "GPU: the supplementary 'pipelined' object, never the headline value; 0 or 1 skips it")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=VALUE",
                    help="engine option (mellow_engine_set_option; repeatable): a non-default configuration is printed in the line's "
                         "`engine` object and the line says so -- the library itself reads no environment variable")
    args = ap.parse_args()
    try:
        engine_options = {kv.split("=", 1)[0]: int(kv.split("=", 1)[1], 0) for kv in args.option}
    except (IndexError, ValueError):
        ap.error("--option wants KEY=INTEGER")
    if args.preset == "configs2":
        args.batch, args.max_len = 32, 300
    elif args.preset == "configs3":
        args.batch, args.max_len, args.clip_seconds = 64, 128, 30
    elif args.preset == "configs4":
        args.batch, args.max_len, args.precision = 128, 64, "fp8"
    # The release library reads no environment variable (round 6: every switch is an engine option, reported by
    # mellow_engine_describe and printed below as `engine`).  What the process environment can still influence is listed in the
    # line: the binding's / this script's own MELLOW_* variables and the HIP runtime's queue count.  Developer probes (MELLOW_DEV_*,
    # honoured only by -DMELLOW_DEVPROBE builds) and the library switches of earlier rounds are refused outright: a number
    # measured under a probe is not a measurement, and a stale switch means the caller expects a configuration it is not getting.
    HARNESS_ENV = {"MELLOW_HIP_LIB", "MELLOW_PRECISION", "MELLOW_BENCH_BACKEND", "MELLOW_BENCH_DEVICE", "MELLOW_BENCH_FORCE_DIST",
                   "MELLOW_DP_AGREE_TIMEOUT_S", "MELLOW_HWQ_SET_BY_IMPORT", "MELLOW_CKPT_DIR", "MELLOW_TOKENIZER_DIR", "MELLOW_DATA_PARALLEL",
                   "MELLOW_DEVICE_RESAMPLE"}
    mellow_env = {k: v for k, v in sorted(os.environ.items()) if k.startswith("MELLOW_") or k == "GPU_MAX_HW_QUEUES"}
    bad = [k for k in mellow_env if k.startswith("MELLOW_") and k not in HARNESS_ENV]
    if bad:
        print(f"bench.py: refusing to run with {bad} set: the library ignores environment variables; use --option KEY=VALUE "
              f"(engine options) instead", file=sys.stderr)
        sys.exit(2)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    # developer overrides used to exercise the multi-process plumbing on a 1-GPU box (2 ranks sharing GPU 0 over gloo);
    # the driver's runs use neither: one rank per GPU, backend "nccl" (= RCCL over xGMI)
    backend = os.environ.get("MELLOW_BENCH_BACKEND", "nccl")
    if "MELLOW_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["MELLOW_BENCH_DEVICE"])
    # MELLOW_BENCH_FORCE_DIST=1 (tests): run the distributed plumbing -- process-group init on "nccl" with device_id, the barrier,
    # the MAX all-reduce of the time, the all-gather of the tokens -- also when the launcher gave this process WORLD_SIZE=1: on a
    # 1-GPU box that is the only way to execute the exact calls the 8-GPU run makes on RCCL
    use_dist = world > 1 or (os.environ.get("MELLOW_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend=backend)
    n_gpus = world if world > 1 else 1
    if args.gpus != n_gpus and rank == 0:
        print(f"# note: --gpus {args.gpus} but WORLD_SIZE={world}; running on {n_gpus} GPU(s)", file=sys.stderr)

    from mellow_amd import synth, dist as mdist
    from mellow_amd.engine import Engine
    dev = local_rank if use_dist else 0
    eng = Engine(device=dev, precision=args.precision, options=engine_options)     # raises if libmellow_hip.so or the GPU is missing
    eng.load_state_dict(synth.make_state_dict(0))
    comm_dev = eng.tdev if backend == "nccl" else torch.device("cpu")      # where the collectives' buffers live
    B, L = args.batch, args.max_len
    from mellow_amd import spec as mspec
    a1, a2, ids = synth.make_batch(B, first=rank * B, n_samples=args.clip_seconds * mspec.SAMPLE_RATE)
    a1d, a2d, idsd = eng._f32(a1), eng._f32(a2), eng._i32(ids)
    n_crops = 1 if args.clip_seconds <= 10 else 7

    def step():
        toks, lens, steps, ftm = eng.generate(a1d, a2d, idsd, max_len=L, top_p=0.8, temperature=1.0, stop_id=0,
                                              ignore_stop=True)
        if use_dist:   # the path's single exchange: all-gather of the token ids over RCCL/xGMI
            mdist.gather_tokens(toks, lens, world * B, L, device=comm_dev)
        return ftm

    _progress("engine ready; warm-up")
    for _ in range(args.warmup):
        step()
    _progress("timed region")
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ftms = [step() for _ in range(args.steps)]
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    phases = eng.last_phase_ms()

    # ---- reference-semantics mode (SURVEY 8d ii): the stop id is honoured, the loop ends when every row has produced it
    #      (reference wrapper.py:247-249).  With the synthetic checkpoint no row emits id 0, so all max_len steps run plus the
    #      host-side check every 8 steps; reported next to the fixed-length headline, never instead of it.
    _progress("leg: ref_sem")
    ref_sem = None
    if rank == 0 and world == 1:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_ref = max(1, min(3, args.steps))
        steps_run = 0
        for _ in range(n_ref):
            _, _, steps_run, _ = eng.generate(a1d, a2d, idsd, max_len=L, top_p=0.8, temperature=1.0, stop_id=0, ignore_stop=False)
        torch.cuda.synchronize()
        ref_sem = {"value": round(n_ref * B / (time.perf_counter() - t1), 2), "unit": "responses/s", "passes": n_ref,
                   "steps_run": int(steps_run), "steps_enqueued": eng.last_steps_enqueued(), "note": "stop id honoured (reference loop exit rule); synthetic weights never emit it"}

    # ---- PCIe-inclusive rate: the same pass with the waveforms and prompt ids starting in (pageable) host memory ----
    _progress("leg: pcie")
    pcie = None
    if rank == 0 and world == 1:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_p = max(1, min(3, args.steps))
        ft_host = []
        for _ in range(n_p):
            eng.generate(a1, a2, ids, max_len=L, top_p=0.8, temperature=1.0, stop_id=0, ignore_stop=True)
            ft_host.append(eng.last_first_token_host_ms)
        torch.cuda.synchronize()
        pcie = {"value": round(n_p * B / (time.perf_counter() - t1), 2), "unit": "responses/s", "passes": n_p,
                "first_token_ms_p50": round(statistics.median(ft_host), 2),
                "note": f"inputs start in pageable HOST memory: host->device copy of 2 x {B} x 1.28 MB waveforms + ids inside the "
                        f"timed region (never the headline); first_token_ms_p50 here is SURVEY 8d's latency definition"}

    # ---- roofline of the dominant kernel family, HIP events on the engine's stream over one more step ----
    _progress("leg: eng.prof_enable(True)")
    eng.prof_enable(True)
    eng.prof_reset()
    step()
    rep = eng.prof_report()
    eng.prof_enable(False)

    if rank == 0:
        traffic, traffic_note = committed_pmc(PMC_TRAFFIC_FILE.replace(".json", "_fp8.json") if args.precision == "fp8" else PMC_TRAFFIC_FILE,
                                              args.precision, "traffic_bytes_per_launch")
        dec_traffic, dec_traffic_note = committed_pmc(PMC_DECODE_FILE, args.precision, "traffic_bytes_per_step")
        total = n_gpus * B * args.steps
        value = total / elapsed
        ms_per_step = elapsed / args.steps * 1e3
        g = rep["gemm_f32_mfma"]
        # ALGORITHMIC flops: the STFT is priced as the FFT it could be (0.051 GF/clip), not as the dense DFT
        # GEMM the kernel runs (2 x frames x 1024 x 1026 flops per clip); the dense figure is kept beside it
        # (when the engine runs the STFT as an FFT -- f32x3 mode -- its own count already is the algorithmic one)
        dense_dft = 0.0 if eng.stft_is_fft() else 2.0 * (2 * B) * 1001 * 1024 * 1026
        flops_alg = g["flops"] - dense_dft + ((2 * B) * FFT_GFLOP_PER_CLIP * 1e9 if dense_dft else 0.0)
        tf = flops_alg / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
        tf_dense = g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
        dec_bytes = decode_algorithmic_bytes(B, L)
        if args.precision == "fp8":
            # the fp8 mode is priced on ITS OWN bytes (e4m3 weights 134.5 MB per step; K/V as the mode stores them):
            # pricing it on the fp32 mode's 538 MB + 46,080 B per token would flatter it
            kv16 = bool(eng.describe()["options"]["fp8_kv16"]["value"])
            dec_bytes = decode_algorithmic_bytes_fp8(B, L) if kv16 else decode_algorithmic_bytes(B, L) - (L - 1) * 134_515_008 * 3.0
        dec_gbs = dec_bytes / (phases["decode_ms"] * 1e-3) / 1e9 if phases["decode_ms"] > 0 else 0.0
        # whole-path two-phase roofline: t_roof = F_dense/P_mfma + Bytes_decode/BW_hbm
        fp8 = args.precision == "fp8"
        peak = PEAK_FP8_MFMA_TFLOPS if fp8 else PEAK_F32X3_TFLOPS if args.precision == "f32x3" else PEAK_F32_MFMA_TFLOPS
        t_roof = dense_gflop_per_response(args.clip_seconds) / (peak * 1e3) * B + dec_bytes / (PEAK_HBM_GBS * 1e9)
        out = {
            "metric": f"audio-pair responses/sec (v0 167M, 2x{args.clip_seconds}s clips, max_len={L}, greedy)" +
                      (" [fp8 e4m3 GEMMs, BASELINE config 5 numerics: NOT the fp32 headline]" if fp8 else "") +
                      (" [exact-fp32-MFMA mode]" if args.precision == "f32" else ""),
            "value": round(value, 2), "unit": "responses/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("fp8_e4m3 GEMMs (Swin linears + LM prefill), f32 accumulate; decode/front-end f32" if fp8 else
                      "f32 (encoder/prefill GEMM operands split EXACTLY into 3 bf16 terms, 6 bf16 MFMA products per fp32 product, "
                      "f32 accumulate: fp32-accurate, parity suite green; mel included, the STFT is an fp32 FFT; attentions and decode "
                      "on exact fp32 MFMA / VALU)" if args.precision == "f32x3" else "f32"),
            "data": "synthetic",
            "config": {"workload": f"v0 167M, batch {B}/GPU, 2x{args.clip_seconds}s 32kHz synthetic clips"
                                   f"{' (7 encoder crops per clip)' if n_crops > 1 else ''} + 16-token prompts, max_len={L}, "
                                   f"greedy, fixed-length (stop id ignored), seeded synthetic weights (real state_dict layout)",
                       "global_batch": n_gpus * B, "max_len": L, "parallelism": f"dp{n_gpus}", "preset": args.preset or "configs1"},
            "ranks_seen": (dist.get_world_size() if use_dist else 1),
            "dist_backend": (dist.get_backend() if use_dist else None),
            "parity_note": "greedy tokens vs the imported reference (seeded synthetic weights): 2,048 / 2,048 over configs[1] (32 x 64, "
                           "tests/golden/b32.npz); 9,453 / 9,600 over configs[2]'s per-rank run (32 x 300, b32long.npz): row 10 takes the other "
                           "side of the reference's own near-tie at step 153 (top-2 gap 8.4e-5, below fp32 summation-order noise) in both "
                           "precision modes and continues on another sequence (147 tokens); the other 31 rows are equal for all 300 steps; 9 "
                           "of the 9,600 reference decisions have a gap < 6e-3 (counted in tests/test_oracle_golden.py)",
            "env": mellow_env,
            "engine": eng.describe(),                  # the library's resolved configuration (mellow_engine_describe); non_default == [] in the driver's run
            "prefill_parts": eng.prefill_parts(),      # measured by the engine: 2 = two half-batch chains on streams that overlap
            "first_token_ms_p50": round(statistics.median(ftms), 2),
            "phase_ms": {k: round(v, 2) for k, v in phases.items()},
            "roofline_gemm": {"kernel": ("gemm_mx8_kernel (v_mfma_scale_f32_32x32x64_f8f6f4: the block-scaled K = 64 form, the pipe the 5 PF peak refers to; MXFP8 activations emitted by their producers, e4m3 weights per output channel) + the standalone quantiser where no producer emits AMX + the fp32 GEMMs left (front-end mel, lm_forward head)"
                                    if fp8 else "gemm_x3q_kernel (LM prefill, Swin stages 2-3) + gemm_x3w_kernel (stage-0 qkv / fc1) + gemm_x3p_kernel (rest of the encoder): 6 x v_mfma_f32_32x32x16_bf16 per fp32 product; "
                                    "mel on gemm_x3p_kernel too, the STFT as an FFT (stft_fft_power_kernel, counted in this family); peak = 2.5 PF dense bf16 / 6" if args.precision == "f32x3"
                                    else "gemm_f32_kernel (v_mfma_f32_32x32x2_f32: encoder + LM prefill GEMMs)"),
                         "bound": "mfma", "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(tf / peak, 4), "traffic": traffic,
                         "traffic_unit": traffic_note,
                         "launches": g["launches"], "avg_launch_us": round(g["ms"] * 1e3 / max(1, g["launches"]), 2),
                         "flops_per_step": flops_alg,
                         "profiled_with_single_chain": True,
                         "profiled_schedule_note": "the per-family HIP events need one launch at a time: this figure comes from one extra pass with the "
                                                   "LM prefill as ONE chain (the timed passes of `value` run it as `prefill_parts` half-batch chains "
                                                   "on overlapping streams, which shortens the prefill phase but not a kernel's own time)",
                         "achieved_dense_dft": round(tf_dense, 2),
                         "gemm_plus_norm_ms": round(g["ms"] + rep["norm"]["ms"], 3),
                         "gemm_plus_norm_note": "family time + the normalisation launches beside it: since round 4 the LM prefill's RMSNorms live in "
                                                "the GEMM epilogues (round 3: 19.6 + 2.37 ms), so this sum is the like-for-like figure across rounds",
                         "note": "achieved = algorithmic flops (STFT priced as an FFT, SURVEY 8d) / family time measured live with "
                                 "HIP events on the engine's stream; achieved_dense_dft counts the DFT GEMM's own flops"},
            "roofline": None,
            "path_roofline": {"t_roof_ms_per_step": round(t_roof * 1e3, 3),
                              "frac": round(t_roof * 1e3 / ms_per_step, 4),
                              "definition": "F_dense/P_mfma(dtype) + Bytes_decode/8TB/s per response x batch (SURVEY 8d)"},
            "kernel_families_ms": {k: round(v["ms"], 3) for k, v in rep.items()},
        }
        # the dominant part of the pass by time is the decode phase (hipGraph replays of one step = one "launch" of 124 kernels):
        # HBM-bound, priced by the ALGORITHMIC bytes of SURVEY 8d; `traffic` = what the memory side actually moved per step (PMC)
        out["roofline"] = {
            "kernel": "one decode step (hipGraph replay): dec_qkv + 30 x dec_attn + 30 x (dec_oproj | dec_gateup16) + 29 x dec_qkv2 + dec_down "
                      "+ final norm, lm_head (f32x3 mode: dec_head3r_kernel, weights streamed once on the bf16 pipe), arg-max; the dominant phase of the pass by time",
            "bound": "hbm", "achieved": round(dec_gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(dec_gbs / PEAK_HBM_GBS, 4), "traffic": dec_traffic, "traffic_unit": dec_traffic_note + " per step",
            "launches": L - 1, "avg_launch_us": round(phases["decode_ms"] * 1e3 / max(1, L - 1), 2),
            "bytes_per_launch": round(dec_bytes / max(1, L - 1)),
            "share_of_pass": round(phases["decode_ms"] / ms_per_step, 3),
            "note": ("achieved = algorithmic bytes per step OF THE fp8 MODE (134.5 MB of e4m3 weights + 23,040 B of bf16 shadow pages per cached token "
                     "per example; fp32 pages with --option fp8_kv16=0) " if args.precision == "fp8" else
                     "achieved = algorithmic bytes per step (538.06 MB of fp32 weights + 46,080 B per cached token per example, SURVEY 8d) ") +
                    "/ average step time, HIP events around the decode phase of the timed pass on the engine's stream",
        }
        out["roofline_decode"] = out["roofline"]                 # the name earlier rounds used
        if ref_sem is not None:
            out["reference_semantics"] = ref_sem
        if pcie is not None:
            out["pcie_inclusive"] = pcie
        std = args.clip_seconds == 10        # the supplementary legs below are defined on the 10 s workload
        if n_gpus == 1 and args.precision != "fp8" and not args.no_alt_modes and std:
            _progress("leg: alt_modes")
            out["alt_modes"] = alt_modes(B, L, args.precision)
        if n_gpus == 1 and args.inflight > 1 and std:
            out["pipelined"] = pipelined(args.inflight, B, L, max(args.steps, 2 * args.inflight), args.precision)
        if n_gpus == 1 and args.precision != "fp8" and not args.no_b64 and B != 64 and std:
            _progress("leg: north_star_b64")
            out["north_star_b64"] = north_star_b64(L, args.precision)
        if n_gpus == 1 and args.precision != "fp8" and not args.no_configs3 and args.preset != "configs3":
            _progress("leg: configs3")
            out["configs3"] = configs3_leg(args.precision)
        if n_gpus == 1 and not args.no_cpu_baseline:
            _progress("leg: cpu_baseline")
            out["cpu_baseline"] = cpu_baseline(L)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
