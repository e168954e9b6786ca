"""GPU (-m gpu): the data-parallel path with REAL engines (BASELINE configs[2]): one process per rank, every
rank with its own engine + weight replica, contiguous shards, ONE gather of the token ids at the end.

The GPU box has one MI355X, so the two ranks share device 0 and rendezvous over gloo (the collective's buffers live on the
host); on an 8-GPU node the same code runs one rank per GPU over RCCL (backend "nccl").  What is exercised here is
everything except the transport: sharding, per-rank engines, ragged/empty shards, early-stopping shards, gather order."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
golden = os.path.join(sys.argv[1], "tests", "golden")
from mellow_amd import dist as mdist, synth
from mellow_amd.engine import Engine
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()

# every collective entry point of torch.distributed, counted: a data-parallel generate may make exactly ONE (north_star)
_COLL = ("all_gather", "all_gather_into_tensor", "all_gather_object", "all_reduce", "broadcast", "broadcast_object_list", "gather",
         "scatter", "reduce", "reduce_scatter", "reduce_scatter_tensor", "all_to_all", "all_to_all_single", "barrier", "send", "recv")
_calls = []
def _count(name, fn):
    def w(*a, **k):
        _calls.append(name)
        return fn(*a, **k)
    return w
for _n in _COLL:
    if hasattr(dist, _n):
        setattr(dist, _n, _count(_n, getattr(dist, _n)))
sd = synth.make_state_dict(0)
eng = Engine(device=0)
eng.load_state_dict(sd)

# (1) fixed-length, B = 5 over 2 ranks (shards 3 + 2): gathered == the same examples in ONE process, and rows 0-1 == goldens
B, L = 5, 8
a1, a2, ids = synth.make_batch(B)
toks, lens = mdist.generate_sharded(eng.generate, a1, a2, ids, max_len=L, stop_id=0, ignore_stop=True)
want, *_ = eng.generate(a1, a2, ids, max_len=L, stop_id=0, ignore_stop=True)
assert toks.shape == (B, L) and np.array_equal(toks, want), (rank, toks, want)
g = np.load(os.path.join(golden, "gen.npz"))
assert np.array_equal(toks[:2], g["tokens"][:, :L])

# (2) reference stop rule per shard: rows (1, 2 | 4) with the mixed-EOS golden's stop id -> rank 0 stops after step 17,
#     rank 1 after step 3; the gathered texts are the reference's
e = np.load(os.path.join(golden, "eos_mixed.npz"))
rows = e["all_stop_examples"].tolist()
a1, a2, ids = synth.make_examples(rows)
toks, lens = mdist.generate_sharded(eng.generate, a1, a2, ids, max_len=int(e["max_len"]), stop_id=int(e["stop_id"]))
assert lens.tolist() == e["all_stop_len"].tolist(), (rank, lens)
for r in range(len(rows)):
    assert toks[r, : lens[r]].tolist() == e[f"all_stop_row{r}"].tolist()

# (3) more ranks than examples: rank 1 gets an empty shard and still takes part in the gather
a1, a2, ids = synth.make_batch(1)
toks, lens = mdist.generate_sharded(eng.generate, a1, a2, ids, max_len=4, stop_id=0, ignore_stop=True)
assert np.array_equal(toks, g["tokens"][:1, :4])

# (4) the public API: MellowWrapper.generate shards by itself under an initialised process group
from mellow_amd import MellowWrapper
class Tok:
    def encode(self, s): return [0] if s == "<|endoftext|>" else [17 + (sum(s.encode()) * 7919 + i * 104729) % 49000 for i, _ in enumerate(s.split())]
    def encode_plus(self, text, max_length=129, **kw):
        ids = self.encode(text)[:max_length]
        return {"input_ids": torch.tensor([ids + [1] * (max_length - len(ids))]), "attention_mask": torch.tensor([[1] * max_length])}
    def decode(self, ids): return " ".join("<|endoftext|>" if int(i) == 0 else f"t{int(i)}" for i in ids)
import wave
def wav(path, f, secs, sr):
    t = np.arange(int(secs * sr)) / sr
    pcm = (0.3 * np.sin(2 * np.pi * f * t) * 32767).astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr); w.writeframes(pcm.tobytes())
tmp = sys.argv[2]
paths = []
for i, (f, secs, sr) in enumerate(((440, 2.5, 44100), (1000, 10.0, 32000), (250, 4.0, 48000))):
    p = os.path.join(tmp, f"clip_{i}.wav")          # one list of paths for every rank (rank 0 writes the files)
    if rank == 0:
        wav(p, f, secs, sr)
    paths.append(p)
dist.barrier()
examples = [[paths[0], paths[1], "compare the two"], [paths[1], paths[2], "which is higher"], [paths[2], paths[0], "describe"]]
m = MellowWrapper(config="v0", model="v0", device=0, use_cuda=True, state_dict=sd, tokenizer=Tok(), data_parallel=True)
del _calls[:]
sharded = m.generate(examples=examples, max_len=6, top_p=0.8, temperature=1.0)
assert _calls == ["all_gather"], _calls        # ONE collective per data-parallel generate: the token gather (the same-examples check uses the store)
# sharding is opt-in and checked: ranks that hold DIFFERENT example lists are refused on every rank, nothing hangs
try:
    m.generate(examples=examples[: 2 + rank], max_len=6, top_p=0.8, temperature=1.0)
    raise SystemExit("mismatching example lists were accepted")
except ValueError as e:
    assert "different `examples`" in str(e), e
# default (like the reference): no sharding, every process answers its own list
m0 = MellowWrapper(config="v0", model="v0", device=0, use_cuda=True, state_dict=sd, tokenizer=Tok())
own = m0.generate(examples=examples[rank: rank + 2], max_len=6, top_p=0.8, temperature=1.0)
assert len(own) == 2
m1 = MellowWrapper(config="v0", model="v0", device=0, use_cuda=True, state_dict=sd, tokenizer=Tok(), data_parallel=False)
alone = m1.generate(examples=examples, max_len=6, top_p=0.8, temperature=1.0)
assert len(sharded) == 3 and sharded == alone, (rank, sharded, alone)
assert own == alone[rank: rank + 2], (rank, own, alone)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def _torchrun(args, port, extra_env=None, timeout=1500, nproc=2):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS=str(max(1, 16 // nproc)),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + args
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def test_two_ranks_real_engines_shard_and_gather(tmp_path):
    script = tmp_path / "dp_worker.py"
    script.write_text(_WORKER)
    r = _torchrun([str(script), ROOT, str(tmp_path)], 29741)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-4000:]
    assert r.stdout.count("ok") == 2


def test_bench_two_ranks_prints_one_line_with_n_gpus_2():
    """The driver's N > 1 launch line (torch.distributed.run, one rank per GPU) with both ranks on GPU 0 over gloo."""
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8"], 29743,
                  {"MELLOW_BENCH_BACKEND": "gloo", "MELLOW_BENCH_DEVICE": "0"})
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["global_batch"] == 16 and out["config"]["parallelism"] == "dp2"


_RCCL_WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from mellow_amd import dist as mdist
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda:0"))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
toks = (np.arange(3 * 5, dtype=np.int32).reshape(3, 5) * 7) % 1000
lens = np.asarray([5, 2, 4], dtype=np.int32)
got, glens = mdist.gather_tokens(toks, lens, 3, 8, device=torch.device("cuda:0"))       # dist.all_gather of cuda blocks = RCCL
assert got.shape == (3, 8) and np.array_equal(got[:, :5], toks) and (got[:, 5:] == -1).all() and glens.tolist() == lens.tolist()
dist.barrier()
dist.destroy_process_group()
print("rccl ok")
"""


def test_gather_runs_on_rccl_with_one_rank(tmp_path):
    """The GPU box has one MI355X, so RCCL cannot be given two ranks -- but the gather's collective (`dist.all_gather` of device
    blocks, the ONE code path of mellow_amd.dist.gather_tokens) can be executed on the "nccl" backend with a single rank: RCCL is
    initialised, the blocks live on the GPU and the call the 8-GPU run makes is the call made here."""
    script = tmp_path / "rccl_worker.py"
    script.write_text(_RCCL_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29747", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29747", str(script), ROOT]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "rccl ok" in r.stdout


def test_bench_runs_end_to_end_on_rccl_with_one_rank():
    """8-GPU readiness on a 1-GPU box: `bench.py` under the driver's launcher (`torch.distributed.run`) with the DEFAULT backend
    ("nccl" = RCCL) and one rank -- process-group init with `device_id`, the barrier, the MAX all-reduce of the elapsed time and
    the token all-gather of mellow_amd.dist all execute on RCCL with device buffers, and `ranks_seen` comes from
    `dist.get_world_size()`.  (MELLOW_BENCH_FORCE_DIST=1: the launcher gives WORLD_SIZE=1, for which bench.py would otherwise
    skip the group.)  The supplementary legs of the N = 1 line are switched off to keep the run short."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29751", HSA_ENABLE_IPC_MODE_LEGACY="0", MELLOW_BENCH_FORCE_DIST="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29751", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "8",
           "--no-cpu-baseline", "--no-alt-modes", "--no-b64", "--no-configs3", "--inflight", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["ranks_seen"] == 1 and out["dist_backend"] == "nccl" and out["value"] > 0


_LOAD_WORKER = r"""
import os, sys, time
sys.path.insert(0, sys.argv[1])
import torch
from mellow_amd import synth
from mellow_amd.engine import Engine
sd = synth.make_state_dict(0)
t0 = time.perf_counter()
eng = Engine(device=0)
eng.load_state_dict(sd)
torch.cuda.synchronize()
t1 = time.perf_counter()
free, total = torch.cuda.mem_get_info(0)
a1, a2, ids = synth.make_batch(4)
toks, *_ = eng.generate(a1, a2, ids, max_len=4, stop_id=0, ignore_stop=True)
print("LOADED %.2f s  used_GiB %.2f  tokens %s" % (t1 - t0, (total - free) / 2**30, toks[0].tolist()), flush=True)
"""


def test_two_replicas_load_concurrently_and_fit(tmp_path):
    """Startup of a data-parallel node, as far as one GPU can show it: two PROCESSES create an engine, load the 479-tensor
    checkpoint (compose + pack kernels included) and finalize AT THE SAME TIME on one device, then both answer.  Checks that
    replica + workspaces of N ranks fit (2 x < 8 GiB here against 288 GB per GPU on the node) and that concurrent loading
    neither fails nor changes the tokens."""
    script = tmp_path / "load_worker.py"
    script.write_text(_LOAD_WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    ps = [subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT)
          for _ in range(2)]
    outs = [p.communicate(timeout=1200) for p in ps]
    for p, (so, se) in zip(ps, outs):
        assert p.returncode == 0, so[-2000:] + se[-4000:]
    lines = [next(l for l in so.splitlines() if l.startswith("LOADED")) for so, _ in outs]
    toks = [l.split("tokens")[1] for l in lines]
    assert toks[0] == toks[1], lines
    for l in lines:
        secs, gib = float(l.split()[1]), float(l.split()[4])
        assert secs < 300 and gib < 16, l
    print(lines)


# ---- eight ranks (BASELINE configs[2] / [4]: 8 x MI355X) on the one GPU of this box, over gloo ------------------------------------
_WORKER8 = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from mellow_amd import dist as mdist, synth, MellowWrapper
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
assert world == 8
_COLL = ("all_gather", "all_gather_into_tensor", "all_gather_object", "all_reduce", "broadcast", "broadcast_object_list", "gather",
         "scatter", "reduce", "reduce_scatter", "reduce_scatter_tensor", "all_to_all", "all_to_all_single", "barrier", "send", "recv")
_calls = []
def _count(name, fn):
    def w(*a, **k):
        _calls.append(name)
        return fn(*a, **k)
    return w
for _n in _COLL:
    if hasattr(dist, _n):
        setattr(dist, _n, _count(_n, getattr(dist, _n)))
class Tok:
    def encode(self, s): return [0] if s == "<|endoftext|>" else [17 + (sum(s.encode()) * 7919 + i * 104729) % 49000 for i, _ in enumerate(s.split())]
    def encode_plus(self, text, max_length=129, **kw):
        ids = self.encode(text)[:max_length]
        return {"input_ids": torch.tensor([ids + [1] * (max_length - len(ids))]), "attention_mask": torch.tensor([[1] * max_length])}
    def decode(self, ids): return " ".join("<|endoftext|>" if int(i) == 0 else f"t{int(i)}" for i in ids)
sd = synth.make_state_dict(0)
# eight replicas load AT ONCE on one device (8 x (weights + workspaces) of 288 GB), eight host polling threads under the cgroup
m = MellowWrapper(config="v0", model="v0", device=0, use_cuda=True, state_dict=sd, tokenizer=Tok(), data_parallel=True, max_positions=512)
# 19 in-memory examples over 8 ranks: shards of ceil(19 / 8) = 3 -> ranks 0..5 hold 3, rank 6 holds 1, rank 7 holds NONE
n = 19
a1, a2, _ = synth.make_batch(n, n_samples=2 * 32000)
examples = [[a1[i], a2[i], f"question number {i} about the two clips"] for i in range(n)]
dist.barrier()                                     # eight replicas finish loading minutes apart under a 16-CPU quota: line the ranks up first
del _calls[:]
got = m.generate(examples=examples, max_len=5, top_p=0.8, temperature=1.0)
assert _calls == ["all_gather"], _calls            # the same-examples agreement of the 8 ranks runs over the store, not a collective
assert len(got) == n
# every rank holds the full, ordered answer list, and it equals the un-sharded answers (computed by rank 0 alone, sent by the store)
store = dist.distributed_c10d._get_default_store()
if rank == 0:
    alone = MellowWrapper(config="v0", model="v0", device=0, use_cuda=True, state_dict=sd, tokenizer=Tok(), data_parallel=False, max_positions=512)
    want = alone.generate(examples=examples, max_len=5, top_p=0.8, temperature=1.0)
    store.set("want8", "\x1f".join(want))
want = store.get("want8").decode().split("\x1f")
assert got == want, (rank, got[:3], want[:3])
try:                                               # one rank out of eight with another list: refused on all eight
    m.generate(examples=examples[: n - (1 if rank == 5 else 0)], max_len=5, top_p=0.8, temperature=1.0)
    raise SystemExit("mismatching example lists were accepted")
except ValueError as e:
    assert "different `examples`" in str(e), e
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "of 8 ok")
"""


def test_eight_ranks_shard_agree_and_gather_on_one_gpu(tmp_path):
    """8-GPU readiness without an 8-GPU node: eight processes (the launcher line of the driver, gloo instead of RCCL, all on GPU
    0) each create a wrapper + engine replica at the same time, agree on the example list through the rendezvous store with eight
    participants, answer ragged shards (3, 3, 3, 3, 3, 3, 1, 0 examples) and take part in exactly ONE all_gather."""
    script = tmp_path / "dp_worker8.py"
    script.write_text(_WORKER8)
    r = _torchrun([str(script), ROOT], 29761, {"MELLOW_DP_AGREE_TIMEOUT_S": "600"}, nproc=8, timeout=2400)
    noise = ("[Gloo]", "closing signal", "error_file", "traceback :", "exitcode", "rank      :", "host      :", "time      :", "------", "======")
    assert r.returncode == 0, "\n".join(l for l in (r.stdout + r.stderr).splitlines() if l.strip() and not any(n in l for n in noise))[-6000:]
    assert r.stdout.count("of 8 ok") == 8


def test_bench_eight_ranks_prints_one_line_with_n_gpus_8():
    """The driver's `--gpus 8` launch line with all ranks on GPU 0 over gloo: one JSON line, `ranks_seen` 8 from
    `dist.get_world_size()`, global batch 8 x 4, weak scaling, the library at its default configuration on every rank."""
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--batch", "4"], 29763,
                  {"MELLOW_BENCH_BACKEND": "gloo", "MELLOW_BENCH_DEVICE": "0"}, timeout=2400, nproc=8)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["ranks_seen"] == 8 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["global_batch"] == 32 and out["config"]["parallelism"] == "dp8"
    assert out["engine"]["non_default"] == [] and out["engine"]["reads_environment"] is False
