#!/usr/bin/env python3
"""Developer probe (GPU box): throughput and phase times of one generate() call at B = 32 / 64 / 128 (max_len 64,
fixed-length), checking that the first 32 rows are bit-identical whatever the batch size."""
import sys, time; sys.path.insert(0, ".")
from _opts import engine_options  # noqa: E402  (--opt KEY=VALUE -> engine options)
OPTS = engine_options()
import numpy as np
from mellow_amd import synth
from mellow_amd.engine import Engine
eng = Engine(device=0, max_positions=1024, options=OPTS); eng.load_state_dict(synth.make_state_dict(0))
ref = None
for B in (32, 64, 128):
    a1, a2, ids = synth.make_batch(B)
    a1d, a2d, idsd = eng._f32(a1), eng._f32(a2), eng._i32(ids)
    t, *_ = eng.generate(a1d, a2d, idsd, max_len=64, stop_id=0, ignore_stop=True)
    if ref is None: ref = t
    assert np.array_equal(t[:32], ref), B
    t0 = time.perf_counter()
    for _ in range(3): eng.generate(a1d, a2d, idsd, max_len=64, stop_id=0, ignore_stop=True)
    dt = (time.perf_counter() - t0) / 3
    print(f"B={B}: {dt*1e3:.1f} ms -> {B/dt:.1f} responses/s  phases {eng.last_phase_ms()}", flush=True)
