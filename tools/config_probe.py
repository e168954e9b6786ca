#!/usr/bin/env python3
"""Developer probe (GPU box): the other BASELINE configurations on one GPU, for the record (not bench lines):
C3 per-rank shape (B=32, max_len 300, sampling args), C4 (B=64, 2x30 s clips = 7 encoder crops per clip, max_len 128)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _opts import engine_options  # noqa: E402  (--opt KEY=VALUE -> engine options)
OPTS = engine_options()
from mellow_amd import spec, synth  # noqa: E402
from mellow_amd.engine import Engine  # noqa: E402

eng = Engine(device=0, max_positions=1024, options=OPTS)
eng.load_state_dict(synth.make_state_dict(0))
for name, B, secs, L in (("C3 per-rank: B=32, 10 s, max_len 300", 32, 10, 300), ("C4: B=64, 30 s, max_len 128", 64, 30, 128)):
    a1, a2, ids = synth.make_batch(B, n_samples=secs * spec.SAMPLE_RATE)
    a1d, a2d, idsd = eng._f32(a1), eng._f32(a2), eng._i32(ids)
    eng.generate(a1d, a2d, idsd, max_len=L, top_p=0.8, temperature=1.0, stop_id=0, ignore_stop=True)
    t0 = time.perf_counter()
    n = 2
    for _ in range(n):
        eng.generate(a1d, a2d, idsd, max_len=L, top_p=0.8, temperature=1.0, stop_id=0, ignore_stop=True)
    dt = (time.perf_counter() - t0) / n
    print(f"{name}: {dt * 1e3:.1f} ms per pass -> {B / dt:.1f} responses/s, phases {eng.last_phase_ms()}", flush=True)
