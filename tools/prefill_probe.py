#!/usr/bin/env python3
"""Developer tool (GPU box): encode / prefill phase times of B=32 passes (same-box A/B metric for the GEMM kernels)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _opts import engine_options  # noqa: E402  (--opt KEY=VALUE -> engine options)
OPTS = engine_options()
from mellow_amd import synth  # noqa: E402
from mellow_amd.engine import Engine  # noqa: E402

eng = Engine(device=0, precision=os.environ.get("MELLOW_PRECISION", "f32x3"), options=OPTS)
eng.load_state_dict(synth.make_state_dict(0))
a1, a2, ids = synth.make_batch(32)
a1d, a2d, idsd = eng._f32(a1), eng._f32(a2), eng._i32(ids)
enc, pre = [], []
for _ in range(6):
    eng.generate(a1d, a2d, idsd, max_len=2, stop_id=0, ignore_stop=True)
    p = eng.last_phase_ms()
    enc.append(p["encode_ms"]); pre.append(p["prefill_ms"])
print("encode_ms", [round(x, 2) for x in enc[1:]], "prefill_ms", [round(x, 2) for x in pre[1:]])
