#!/usr/bin/env python3
"""Developer tool (GPU box): BASELINE config 5 -- the fp8 GEMM mode against the exact-fp32 mode on the same inputs:
prefill logits error, greedy token agreement, throughput."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _opts import engine_options  # noqa: E402  (--opt KEY=VALUE -> engine options)
OPTS = engine_options()
from mellow_amd import synth  # noqa: E402
from mellow_amd.engine import Engine  # noqa: E402

STRUCT = len(sys.argv) > 1 and sys.argv[1] == "structured"
sd = synth.make_state_dict(0, structured=STRUCT)
print("checkpoint:", "structured (decaying singular spectrum)" if STRUCT else "default (i.i.d. Gaussian)")
e32 = Engine(device=0, max_positions=1024, options=OPTS)
e32.load_state_dict(sd)
e8 = Engine(device=0, max_positions=1024, precision="fp8", options=OPTS)
e8.load_state_dict(sd)
B, L = 32, 64
a1, a2, ids = synth.make_batch(B)
p32 = e32.prefix(a1, a2, ids)
p8 = e8.prefix(a1, a2, ids)
print(f"prefix (encoder in fp8): max|d| {float((p8 - p32).abs().max()):.3e} of max|ref| {float(p32.abs().max()):.3e}; "
      f"rel rms {float((p8 - p32).pow(2).mean().sqrt() / p32.pow(2).mean().sqrt()):.3e}")
l32 = e32.lm_prefill(p32, reserve=2).cpu()
l8 = e8.lm_prefill(p32, reserve=2).cpu()          # same (fp32) prefix: isolates the LM prefill
print(f"prefill logits (fp8 LM on the fp32 prefix): rel rms {float((l8 - l32).pow(2).mean().sqrt() / l32.pow(2).mean().sqrt()):.3e}, "
      f"arg-max agreement {float((l8.argmax(-1) == l32.argmax(-1)).float().mean()):.3f}")
t32, *_ = e32.generate(a1, a2, ids, max_len=L, stop_id=0, ignore_stop=True)
t8, *_ = e8.generate(a1, a2, ids, max_len=L, stop_id=0, ignore_stop=True)
first = float((t8[:, 0] == t32[:, 0]).mean())
prefix_len = [int(np.argmax(np.concatenate([(r8 != r32), [True]]))) for r8, r32 in zip(t8, t32)]
print(f"greedy tokens: first-token agreement {first:.3f}; mean common prefix {np.mean(prefix_len):.1f} of {L} tokens; "
      f"rows identical over all {L}: {float(np.mean([p == L for p in prefix_len])):.3f}; position-wise agreement {float((t8 == t32).mean()):.3f}")
for name, eng in (("f32", e32), ("fp8", e8)):
    a1d, a2d, idsd = eng._f32(a1), eng._f32(a2), eng._i32(ids)
    eng.generate(a1d, a2d, idsd, max_len=L, stop_id=0, ignore_stop=True)
    t0 = time.perf_counter()
    for _ in range(3):
        eng.generate(a1d, a2d, idsd, max_len=L, stop_id=0, ignore_stop=True)
    dt = (time.perf_counter() - t0) / 3
    print(f"{name}: {dt * 1e3:.1f} ms per B={B} pass -> {B / dt:.1f} responses/s, phases {eng.last_phase_ms()}")
