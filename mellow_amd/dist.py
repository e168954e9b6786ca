"""Data-parallel sharding of a batch of (audio1, audio2, prompt) examples over the GPUs of one node
(SURVEY.md §8e): examples are independent, so each rank (one process per GPU) runs the whole hot path on a
contiguous shard with its own weight replica, and the only communication is ONE all-gather of the generated
token ids at the end (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).
The payload is tiny (int32 [B_local, max_len] + lengths), so the collective is latency-bound.  The opt-in sharding of
`MellowWrapper.generate` first checks that every rank holds the same examples -- through the rendezvous store, not through a
second collective (`agree_on_examples`)."""
from __future__ import annotations

from typing import List, Tuple

import os

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of ceil(n/world) examples per rank (the last ranks may be short or empty)."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


_agree_calls = 0


def examples_signature(examples) -> bytes:
    """count + SHA-256 of a list of [audio1, audio2, prompt] examples.  File names and prompts are hashed as text; in-memory
    audio (ndarray / tensor, the wrapper's extension) is hashed by CONTENT -- shape, dtype and the raw bytes of its contiguous
    float32 form -- because `str()` of an array is numpy's summarised form ('[0.1 0.2 ... 0.3]') and two different clips can
    print alike."""
    import hashlib
    h = hashlib.sha256()
    n = 0
    for ex in examples:
        n += 1
        for slot, item in enumerate(ex):
            if isinstance(item, (str, bytes)) or hasattr(item, "__fspath__"):
                b = item if isinstance(item, bytes) else str(item).encode()
                h.update(b"s" + len(b).to_bytes(8, "little") + b)
                # An AUDIO slot (index 0 / 1) that names a file on this rank: its size and a digest of its first and last 64 KiB
                # go in too, so that same-named but different files on different nodes do not pass for the same example.  No
                # mtime (copies of one file on two nodes differ in it), and the prompt slot is never looked up on disk (a prompt
                # that happens to name a file in one rank's working directory is still the same prompt).
                if slot < 2:
                    try:
                        size = os.path.getsize(item)
                        h.update(b"f" + int(size).to_bytes(8, "little"))
                        with open(item, "rb") as f:
                            h.update(f.read(65536))
                            if size > 131072:
                                f.seek(size - 65536)
                                h.update(f.read(65536))
                    except (OSError, ValueError, TypeError):
                        pass
            else:
                a = np.ascontiguousarray(torch.as_tensor(item).detach().cpu().numpy(), dtype=np.float32)
                h.update(b"a" + repr(a.shape).encode() + a.tobytes())
    return n.to_bytes(8, "little") + h.digest()


def agree_on_examples(sig: bytes) -> None:
    """Every rank must have been handed the same `examples` before they are sharded: the signatures are compared through the
    process group's rendezvous STORE (host TCP to the store rank 0 already runs for torch.distributed): NOT a collective, no GPU
    work, no RCCL call -- the token all-gather stays the one collective of a data-parallel `generate` (north_star).  It cannot
    ride in that gather: ranks holding lists of different LENGTH would enter it with blocks of different size (undefined
    behaviour under RCCL), which is exactly the case to refuse.  Raises ValueError on every rank when the lists differ."""
    global _agree_calls
    store = dist.distributed_c10d._get_default_store()
    rank, world = dist.get_rank(), dist.get_world_size()
    _agree_calls += 1
    c = _agree_calls                                     # every rank makes the same sequence of data-parallel calls
    store.set(f"mellow_amd/examples/{c}/{rank}", sig)
    # The key carries this process's COUNT of data-parallel calls: a rank that raised before reaching an earlier call (an audio
    # file missing on one node is enough), or that drives a second wrapper differently, is one call behind for good.  Waiting for
    # it with the store's default timeout (300 s) and an opaque error helps nobody: wait explicitly, briefly, and say what happened.
    import datetime
    timeout_s = float(os.environ.get("MELLOW_DP_AGREE_TIMEOUT_S", "120"))
    keys = [f"mellow_amd/examples/{c}/{r}" for r in range(world)]
    try:
        store.wait(keys, datetime.timedelta(seconds=timeout_s))
    except Exception as e:
        missing = []
        for r, k in enumerate(keys):
            try:
                if not store.check([k]):
                    missing.append(r)
            except Exception:
                missing.append(r)
        raise RuntimeError(f"data_parallel generate(): rank(s) {missing} did not reach their data-parallel call #{c} within {timeout_s:.0f} s "
                           "(every rank must make the same sequence of data-parallel generate() calls; a rank that raised before "
                           "one of them, or skipped it, is out of step -- restart the group).  MELLOW_DP_AGREE_TIMEOUT_S sets the wait.") from e
    sigs = [bytes(store.get(k)) for k in keys]
    if c > 1:          # every rank has published call c, hence finished reading call c - 1
        try:
            store.delete_key(f"mellow_amd/examples/{c - 1}/{rank}")
        except Exception:
            pass
    if any(s != sigs[0] for s in sigs):
        counts = [int.from_bytes(s[:8], "little") for s in sigs]
        raise ValueError("data_parallel generate(): the ranks were given different `examples` "
                         f"(counts {counts}); call it with the same list on every rank, or turn sharding off")


def gather_tokens(tokens: np.ndarray, lengths: np.ndarray, n_total: int, max_len: int, device=None, per_rank: int = 0):
    """All-gather variable-size shards: tokens int32 [n_local, steps_local<=max_len], lengths int32 [n_local].
    Returns (tokens [n_total, max_len] padded with -1, lengths [n_total]) on every rank.
    ONE code path for every backend: `dist.all_gather` of equally sized int32 blocks into a list of tensors on `device`
    (cuda under "nccl" = RCCL over xGMI, cpu under gloo), so the line the 8-GPU run executes is the line the tests executed."""
    if not (dist.is_available() and dist.is_initialized()):
        out = np.full((n_total, max_len), -1, dtype=np.int32)
        out[: tokens.shape[0], : tokens.shape[1]] = tokens
        return out, np.asarray(lengths, dtype=np.int32)
    world = dist.get_world_size()
    per = per_rank or (n_total + world - 1) // world
    dev = device if device is not None else torch.device("cpu")
    buf = torch.full((per, max_len + 1), -1, dtype=torch.int32, device=dev)
    if tokens.shape[0]:
        buf[: tokens.shape[0], : tokens.shape[1]] = torch.as_tensor(tokens, dtype=torch.int32, device=dev)
        buf[: tokens.shape[0], max_len] = torch.as_tensor(lengths, dtype=torch.int32, device=dev)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    out = torch.cat(parts, 0).cpu().numpy()[:n_total]
    return out[:, :max_len].copy(), out[:, max_len].copy()


def generate_sharded(generate_fn, audio1, audio2, input_ids, max_len: int, device=None, **kw):
    """Run `generate_fn(audio1_shard, audio2_shard, ids_shard, max_len=..., **kw) -> (tokens, lengths, steps, ...)`
    on this rank's shard and gather everything on every rank."""
    n = len(audio1)
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = shard_range(n, rank, world)
    if hi > lo:
        res = generate_fn(audio1[lo:hi], audio2[lo:hi], input_ids[lo:hi], max_len=max_len, **kw)
        toks, lens = np.asarray(res[0], dtype=np.int32), np.asarray(res[1], dtype=np.int32)
    else:
        toks, lens = np.zeros((0, 0), dtype=np.int32), np.zeros((0,), dtype=np.int32)
    return gather_tokens(toks, lens, n, max_len, device)
