#!/usr/bin/env python3
"""Developer tool (GPU box): a decode-heavy workload (B=32, max_len=64, 3 passes) for `rocprofv3 --kernel-trace` and for
quick A/B timing of the decode phase: prints the phase times of the last pass."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _opts import engine_options  # noqa: E402  (--opt KEY=VALUE -> engine options)
OPTS = engine_options()
from mellow_amd import synth  # noqa: E402
from mellow_amd.engine import Engine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = int(sys.argv[2]) if len(sys.argv) > 2 else 64
eng = Engine(device=0, precision=os.environ.get("MELLOW_PRECISION", "f32x3"), options=OPTS)
eng.load_state_dict(synth.make_state_dict(0))
a1, a2, ids = synth.make_batch(B)
a1d, a2d, idsd = eng._f32(a1), eng._f32(a2), eng._i32(ids)
dec, enc, pre = [], [], []
for _ in range(4):
    toks, *_ = eng.generate(a1d, a2d, idsd, max_len=L, stop_id=0, ignore_stop=True)
    ph = eng.last_phase_ms()
    dec.append(ph["decode_ms"]); enc.append(ph["encode_ms"]); pre.append(ph["prefill_ms"])
print("decode_ms per pass", [round(d, 2) for d in dec], " ms/step", round(min(dec) / (L - 1), 4), " tokens[0,:6]", toks[0, :6].tolist(),
      " encode_ms", round(min(enc), 2), " prefill_ms", round(min(pre), 2))
