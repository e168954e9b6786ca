#!/usr/bin/env python3
"""Small workload for rocprofv3 --pmc passes: encoder + prefill only (the dense GEMM family) at B=32, in the bench's numeric
mode (MELLOW_PRECISION, default f32x3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _opts import engine_options  # noqa: E402  (--opt KEY=VALUE -> engine options)
OPTS = engine_options()
from mellow_amd import synth
from mellow_amd.engine import Engine
eng = Engine(device=0, precision=os.environ.get("MELLOW_PRECISION", "f32x3"), options=OPTS)
eng.load_state_dict(synth.make_state_dict(0))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
a1, a2, ids = synth.make_batch(B)
a1d, a2d, idsd = eng._f32(a1), eng._f32(a2), eng._i32(ids)
for _ in range(2):
    eng.generate(a1d, a2d, idsd, max_len=1, stop_id=0, ignore_stop=True)
print("done")
