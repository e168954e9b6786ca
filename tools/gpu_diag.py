#!/usr/bin/env python3
"""Developer diagnostic (GPU box): every parity tap of the HIP engine against the oracle, printed as a
table without stopping at the first mismatch, then a timing run.  Not a test; tests/ hold the asserts."""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mellow_amd import spec, synth  # noqa: E402
from mellow_amd.engine import Engine  # noqa: E402
from oracle import mellow_oracle as O  # noqa: E402


def diff(name, got, ref):
    got = torch.as_tensor(got).detach().cpu().double().reshape(-1)
    ref = torch.as_tensor(ref).detach().cpu().double().reshape(-1)
    if got.numel() != ref.numel():
        print(f"  {name:14s} SIZE MISMATCH got {got.numel()} ref {ref.numel()}")
        return
    d = (got - ref).abs()
    bad = int((~torch.isfinite(got)).sum())
    print(f"  {name:14s} max|d| {d.max():.3e}  rel {d.max() / (ref.abs().max() + 1e-30):.3e}  "
          f"mean|d| {d.mean():.3e}  max|ref| {ref.abs().max():.3e}  nonfinite {bad}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-parity", action="store_true")
    ap.add_argument("--bench-b", type=int, default=32)
    ap.add_argument("--max-len", type=int, default=64)
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()
    torch.set_num_threads(min(32, os.cpu_count()))
    t0 = time.time()
    sd = synth.make_state_dict(0)
    eng = Engine(device=0, max_positions=1024)
    eng.load_state_dict(sd)
    print(f"engine loaded ({time.time() - t0:.1f}s)", flush=True)
    if args.no_graph:
        eng.set_graph(False)

    if not args.skip_parity:
        B = 2
        a1, a2, ids = synth.make_batch(B)
        a1t, a2t, idst = torch.from_numpy(a1), torch.from_numpy(a2), torch.from_numpy(ids)
        taps = {}
        with torch.no_grad():
            oprefix = O.generate_prefix_inference(sd, a1t, a2t, idst, taps)
            # second clip set (audio2) for the prefix rows
        print(f"oracle prefix done ({time.time() - t0:.1f}s)", flush=True)
        eng.enable_taps(True)
        lm = eng.logmel(a1, apply_bn=False).cpu()
        diff("logmel", lm, taps["logmel"][:, 0])
        lmb = eng.logmel(a1, apply_bn=True).cpu()
        diff("logmel_bn", lmb, taps["logmel_bn"][:, 0])
        enc = eng.encode(a1).cpu()
        pw = eng.tap("power").cpu().reshape(B, 1001, 544)[:, :, :513]
        diff("power", pw, taps["power"][:, 0])
        diff("patch", eng.tap("patch").cpu().reshape(B, 4096, 96), taps["patch"])
        for s in range(4):
            diff(f"stage{s}", eng.tap(f"stage{s}").cpu(), taps[f"stage{s}"])
        fpx = eng.tap("fpx").cpu().reshape(B, 32, 544)[:, :, :527]
        diff("fpx", fpx, taps["framewise"][:, 0::32])
        emb33 = eng.tap("emb33").cpu().reshape(B, 33, 768)
        ref_emb33 = torch.cat((taps["embedding"][:, :1], taps["embedding"][:, 1::32]), 1)
        diff("latent", emb33[:, 0], taps["latent"])
        diff("emb33", emb33, ref_emb33)
        ref_p33 = torch.cat((taps["projected"][:, :1], taps["projected"][:, 1::32]), 1)
        diff("proj33", eng.tap("proj33").cpu().reshape(B, 33, 576), ref_p33)
        diff("audio129", enc, taps["audio1_ds"])
        pre = eng.prefix(a1, a2, ids).cpu()
        diff("prefix", pre, oprefix)
        eng.enable_taps(False)

        # ---- LM on the ORACLE prefix (isolates LM kernels from encoder error) ----
        gen = np.load(os.path.join(ROOT, "tests", "golden", "gen.npz"))
        with torch.no_grad():
            ol = O.llama_forward(sd, O.LMParams(), oprefix, last_only=True)[:, -1]
        l0 = eng.lm_prefill(oprefix, reserve=32).cpu()
        diff("prefill_logit", l0, ol)
        diff("  vs golden", l0, gen["logits_step0"])
        print("  argmax engine", l0.argmax(-1).tolist(), "golden", gen["tokens"][:, 0].tolist())
        toks = gen["tokens"]
        for i in range(1, min(6, toks.shape[1])):
            li = eng.lm_decode_step(toks[:, i - 1]).cpu()
            diff(f"decode{i}_sub", li[:, gen["sub_vocab"]], gen["logits_sub"][i])
            print("  argmax engine", li.argmax(-1).tolist(), "golden", toks[:, i].tolist())
        # ---- end to end ----
        t, lens, steps, ftm = eng.generate(a1, a2, ids, max_len=12, stop_id=0)
        print("generate tokens:\n", t, "\n golden:\n", gen["tokens"], "\n match:", np.array_equal(t, gen["tokens"]),
              "steps", steps, "first_token_ms %.2f" % ftm, flush=True)

    # ---- timing ----
    B = args.bench_b
    a1, a2, ids = synth.make_batch(B)
    a1d, a2d, idsd = eng._f32(a1), eng._f32(a2), eng._i32(ids)
    for it in range(3):
        t1 = time.time()
        t, lens, steps, ftm = eng.generate(a1d, a2d, idsd, max_len=args.max_len, stop_id=0, ignore_stop=True)
        dt = time.time() - t1
        print(f"generate B={B} L={args.max_len}: {dt * 1e3:.1f} ms  -> {B / dt:.1f} responses/s  first_token {ftm:.1f} ms  "
              f"phases {eng.last_phase_ms()}", flush=True)
    eng.prof_enable(True)
    eng.prof_reset()
    eng.generate(a1d, a2d, idsd, max_len=args.max_len, stop_id=0, ignore_stop=True)
    rep = eng.prof_report()
    eng.prof_enable(False)
    for k, v in rep.items():
        tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0
        gb = v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0
        print(f"  {k:20s} launches {v['launches']:6d}  {v['ms']:9.3f} ms  {tf:8.2f} TFLOP/s  {gb:9.1f} GB/s")


if __name__ == "__main__":
    main()
