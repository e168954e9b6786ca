"""Serving-side pipelining: several engine contexts on ONE MI355X, batches dealt to them round-robin.

Why: a decode step is a chain of ~150 small dependent launches that leaves most of the chip idle, while the
encoder / prefill phases are dense MFMA work.  Two or three independent contexts (own stream, own activation and
KV buffers; ONE shared weight copy: `Engine.fork` / `mellow_engine_fork`) overlap those phases of different batches; bench.py
--inflight N reports the measured rate as its supplementary `pipelined` object.  Latency of a single
batch gets worse (contention), so this is a throughput mode for a serving front-end, not the default path and not
what bench.py reports as `value`.

The reference has no counterpart (its generate() is synchronous, wrapper.py:258-287); results are identical to
calling one engine batch by batch (rows are independent, every context holds the same weights)."""
from __future__ import annotations

import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List, Optional, Sequence

import torch

from .engine import Engine
from .spec import LMConfig


class EnginePool:
    """`n_contexts` engines on one device; `generate_many` pipelines a list of batches over them."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], n_contexts: int = 2, device: int = 0,
                 max_positions: Optional[int] = None, lm: Optional[LMConfig] = None, precision: Optional[str] = None):
        if n_contexts < 1:
            raise ValueError("n_contexts must be >= 1")
        # Every context has its own HIP stream; HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless the variable
        # says otherwise) and reads it ONCE, when the runtime initialises.  `import mellow_amd` sets it to 8 if the host has not --
        # which has no effect when the host process touched the GPU before the import.  With fewer queues than contexts two
        # contexts share a queue and serialise (right answers, `pipelined` 543 -> 443 responses/s measured): say so.
        import os
        import warnings
        q = os.environ.get("GPU_MAX_HW_QUEUES")
        if n_contexts > 1 and (q is None or not q.isdigit() or int(q) < min(8, n_contexts + 1)):
            warnings.warn(f"EnginePool({n_contexts} contexts): GPU_MAX_HW_QUEUES={q!r}; export GPU_MAX_HW_QUEUES=8 before the process's "
                          "first GPU call, or the contexts will share hardware queues and partly serialise")
        if n_contexts > 1 and torch.cuda.is_initialized() and os.environ.get("MELLOW_HWQ_SET_BY_IMPORT") == "1":
            warnings.warn("EnginePool: the GPU was initialised before `import mellow_amd` set GPU_MAX_HW_QUEUES=8, so HIP may still map "
                          "streams onto 4 hardware queues; export the variable in the environment of the process instead")
        first = Engine(lm=lm, device=device, max_positions=max_positions, precision=precision)
        first.load_state_dict(state_dict)
        self.engines: List[Engine] = [first] + [first.fork() for _ in range(n_contexts - 1)]      # one weight arena, N contexts
        self._locks = [threading.Lock() for _ in self.engines]
        self._pool = ThreadPoolExecutor(max_workers=n_contexts)

    def close(self):
        self._pool.shutdown(wait=True)
        for e in reversed(self.engines):      # forks before their parent
            e.close()
        self.engines = []

    def _run(self, slot: int, batch, kw):
        with self._locks[slot]:          # one call at a time per context (the C ABI serialises per handle)
            a1, a2, ids = batch
            return self.engines[slot].generate(a1, a2, ids, **kw)     # ctypes releases the GIL inside the call

    def generate_many(self, batches: Sequence, **kw):
        """batches: sequence of (audio1, audio2, input_ids); returns the per-batch results of Engine.generate, in order."""
        futs = [self._pool.submit(self._run, i % len(self.engines), b, kw) for i, b in enumerate(batches)]
        return [f.result() for f in futs]
