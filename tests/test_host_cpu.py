"""CPU: host logic of the drop-in wrapper — audio ingest (reference wrapper.py:141-168), config / error
conventions, data-parallel sharding + gather over gloo (world_size 2)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from mellow_amd import audio, dist as mdist, spec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_wav(path, x, sr):
    from scipy.io import wavfile
    wavfile.write(path, sr, (np.clip(x, -1, 1) * 32767).astype(np.int16))


def test_tile_short_clip(tmp_path):
    sr = 32000
    x = np.sin(np.arange(100000) * 0.01).astype(np.float32) * 0.5
    p = str(tmp_path / "short.wav")
    _write_wav(p, x, sr)
    y = audio.load_audio_into_tensor(p, 10, sr)
    assert y.shape == (320000,) and y.dtype == torch.float32
    w, _ = audio.load_wav(p)
    assert torch.equal(y[:100000], w[0]) and torch.equal(y[100000:200000], w[0]) and torch.equal(y[300000:], w[0, :20000])


def test_crop_long_clip_with_injected_start(tmp_path):
    sr = 32000
    x = (np.arange(400000) % 1000 / 1000.0 - 0.5).astype(np.float32)
    p = str(tmp_path / "long.wav")
    _write_wav(p, x, sr)
    w, _ = audio.load_wav(p)
    y = audio.load_audio_into_tensor(p, 10, sr, start_index=1234)
    assert torch.equal(y, w[0, 1234:1234 + 320000])
    y2 = audio.load_audio_into_tensor(p, 10, sr)              # random start (unseeded, like the reference)
    assert y2.shape == (320000,)


def test_multichannel_is_flattened_not_mixed(tmp_path):
    from scipy.io import wavfile
    sr = 32000
    st = np.stack([np.full(1000, 1000, np.int16), np.full(1000, -2000, np.int16)], 1)
    p = str(tmp_path / "st.wav")
    wavfile.write(p, sr, st)
    y = audio.load_audio_into_tensor(p, 1, sr)
    assert abs(float(y[0]) - 1000 / 32768) < 1e-7 and abs(float(y[1000]) + 2000 / 32768) < 1e-7   # ch0 then ch1


def test_resample_44k_to_32k_length_and_tone():
    sr0, sr1 = 44100, 32000
    n = 403604                                                # resource/1.wav of the reference
    t = np.arange(n) / sr0
    x = torch.from_numpy(np.sin(2 * np.pi * 440 * t).astype(np.float32))[None]
    y = audio.resample(x, sr0, sr1)
    assert y.shape == (1, 292865)                             # ceil(320 * n / 441)
    ref = np.sin(2 * np.pi * 440 * np.arange(y.shape[1]) / sr1)
    assert np.abs(y[0, 2000:-2000].numpy() - ref[2000:-2000]).max() < 2e-3
    assert audio.resample(x, 32000, 32000) is x               # no-op at the model rate


def test_reference_fixture_wavs_load():
    p = "/root/reference/resource/1.wav"
    if not os.path.exists(p):
        pytest.skip("reference fixtures are only present in the build container")
    y = audio.load_audio_into_tensor(p, 10, 32000)
    assert y.shape == (320000,)
    w, sr = audio.load_wav(p)
    assert sr == 44100 and w.shape == (1, 403604)
    r = audio.resample(w, sr, 32000)
    assert torch.equal(y[: r.shape[1]], r[0]) and torch.equal(y[r.shape[1]:], r[0, : 320000 - r.shape[1]])


def test_host_ingest_against_the_reference_preprocess_audio_goldens():
    """A0 on the committed fixture PCM (tests/golden/example.npz: resource/1.wav, 2.wav decoded to int16 -- data -- and the
    arrays the REFERENCE's own preprocess_audio returned for them, generated in the build container): the tile case
    403,604 @ 44.1 kHz -> 292,865 -> repeated to 320,000, and the crop case 445,940 -> 323,585 cut at the offset the reference
    drew from `random` under the stored seed.  The resampling step of the golden run was the independent fp64 oracle of
    torchaudio's published algorithm (oracle/resample_oracle.py via tests/golden/ref_shims; torchaudio's binary is absent), so
    the product's fp32 resampler is compared with something it shares no code with; decode, flatten, tile, crop and the draw
    are pinned to the reference's own code."""
    import random
    g = np.load(os.path.join(ROOT, "tests", "golden", "example.npz"))
    assert int(g["sr1"]) == 44100 and g["pcm1"].shape == (403604,) and g["pcm2"].shape == (445940,) and g["pcm1"].dtype == np.int16
    w1 = torch.from_numpy(g["pcm1"].astype(np.float32) / 32768.0)[None]
    w2 = torch.from_numpy(g["pcm2"].astype(np.float32) / 32768.0)[None]
    r1, r2 = audio.resample(w1, 44100, 32000)[0], audio.resample(w2, 44100, 32000)[0]
    assert r1.shape == (292865,) and r2.shape == (323585,)
    t1 = audio.fit_duration(r1, 320000)
    assert torch.equal(t1[:292865], r1) and torch.equal(t1[292865:], r1[: 320000 - 292865])
    random.seed(int(g["seed"]))
    t2 = audio.fit_duration(r2, 320000)                               # draws random.randrange(323585 - 320000) like wrapper.py:164
    start = int(g["crop_start2"])
    assert torch.equal(t2, r2[start:start + 320000])
    assert torch.equal(audio.fit_duration(r2, 320000, start_index=start), t2)
    for t, name in ((t1, "audio1"), (t2, "audio2")):
        assert float((t[::61] - torch.from_numpy(g[f"{name}_sub"])).abs().max()) <= 2e-6, name
        assert abs(float(t.double().sum()) - float(g[f"{name}_sum"])) <= 0.5
        assert abs(float(t.double().abs().sum()) - float(g[f"{name}_abs"])) <= 0.5


def test_wrapper_error_conventions():
    from mellow_amd import MellowWrapper
    with pytest.raises(ValueError, match="not supported"):
        MellowWrapper(config="v0", model="v9", device=0)
    assert MellowWrapper.model_repo == "soham97/mellow"
    assert MellowWrapper.model_name == {"v0": "v0.ckpt", "v0_s": "v0_s.ckpt"}
    from mellow_amd.wrapper import get_audio_encoder, get_model_class
    with pytest.raises(NotImplementedError):
        get_model_class("Other")
    with pytest.raises(Exception, match="incorrect or not supported"):
        get_audio_encoder("CNN14")
    # no CPU model path: the product never falls back to eager PyTorch / the oracle
    with pytest.raises(RuntimeError, match="no CPU path"):
        MellowWrapper(config="v0", model="v0", device="cpu", use_cuda=False, state_dict={})


def test_config_keys_match_reference_layout():
    import yaml
    cfg = yaml.safe_load(open(os.path.join(ROOT, "mellow_amd", "config", "v0.yaml")))
    assert cfg["data"] == {"sampling_rate": 32000, "segment_seconds": 10, "tokenizer_type": "HuggingFaceTB/SmolLM2-135M",
                           "text_tokenization_len": 129}
    assert cfg["model"]["encoder"] == {"audioenc_name": "HTSAT", "transformer_embed_dim": 768, "out_emb": 768, "d_proj": 576}
    assert cfg["model"]["decoder"] == {"text_decoder": "HuggingFaceTB/SmolLM2-135M", "prefix_length": 389}
    assert cfg["model"]["model_type"] == "Mellow"
    lm = spec.LMConfig.load()
    assert (lm.vocab_size, lm.hidden_size, lm.num_hidden_layers, lm.num_attention_heads, lm.num_key_value_heads) == \
        (49152, 576, 30, 9, 3)


def test_shard_ranges():
    assert [mdist.shard_range(256, r, 8) for r in range(8)] == [(32 * r, 32 * r + 32) for r in range(8)]
    assert [mdist.shard_range(5, r, 4) for r in range(4)] == [(0, 2), (2, 4), (4, 5), (5, 5)]
    assert mdist.shard_range(1, 0, 1) == (0, 1)


_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from mellow_amd import dist as mdist
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
n, L = 5, 7
def fake_generate(a1, a2, ids, max_len, **kw):
    # token = 100*example + step; ragged step counts per shard, like early-stopping shards
    steps = max_len - rank
    toks = np.stack([100 * int(i) + np.arange(steps) for i in ids]).astype(np.int32).reshape(len(ids), steps)
    return toks, np.full(len(ids), steps - 1, np.int32), steps, 0.0
ex = np.arange(n)
toks, lens = mdist.generate_sharded(fake_generate, ex, ex, ex, max_len=L)
assert toks.shape == (n, L) and lens.shape == (n,)
for i in range(n):
    owner = [r for r in range(world) if mdist.shard_range(n, r, world)[0] <= i < mdist.shard_range(n, r, world)[1]][0]
    steps = L - owner
    assert toks[i, :steps].tolist() == (100 * i + np.arange(steps)).tolist(), (rank, i, toks[i])
    assert (toks[i, steps:] == -1).all() and lens[i] == steps - 1
assert _calls == ["all_gather"], _calls                 # the gather is ONE collective
# the cross-rank check of the opt-in sharding (wrapper._check_same_examples) goes through the rendezvous store: no collective
del _calls[:]
same = [["a.wav", "b.wav", "compare"], ["c.wav", "d.wav", "describe"]]
mdist.agree_on_examples(mdist.examples_signature(same))
for bad in (same[: 1 + rank],                                            # different COUNT per rank
            [["a.wav", "b.wav", "compare"], ["c.wav", "d.wav", "describe" + "!" * rank]]):      # same count, different content
    try:
        mdist.agree_on_examples(mdist.examples_signature(bad))
        raise SystemExit("mismatching example lists were accepted")
    except ValueError as e:
        assert "different `examples`" in str(e), e
# in-memory audio is hashed by content: two long clips whose str() is the same summarised text must not collide
x = np.zeros(5000, np.float32); y = x.copy(); y[2500] = 0.25
assert str(x) == str(y) and mdist.examples_signature([[x, x, "p"]]) != mdist.examples_signature([[y, x, "p"]])
assert mdist.examples_signature([[torch.from_numpy(y), x, "p"]]) == mdist.examples_signature([[y, x, "p"]])
try:
    mdist.agree_on_examples(mdist.examples_signature([[x if rank == 0 else y, x, "p"]]))
    raise SystemExit("different clips with the same printed form were accepted")
except ValueError:
    pass
mdist.agree_on_examples(mdist.examples_signature(same))       # and the sequence stays in step after refusals
assert _calls == [], _calls
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_dp_gather_two_process_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29731", str(script), ROOT]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


def test_bench_cli_contract_flags():
    """bench.py must accept the driver's flags (and the supplementary ones) without touching a GPU: --help exits 0."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--precision", "--inflight", "--no-cpu-baseline", "--preset", "configs3", "--clip-seconds"):
        assert flag in r.stdout


def test_one_default_numeric_mode_everywhere():
    """the mode a user gets with no keyword is the mode the bench line's `dtype` names -- f32x3 in the C
    library's header, the ctypes `Engine`, `MellowWrapper` and bench.py alike (the GPU twin:
    test_gpu_parity.py::test_wrapper_without_keywords_runs_the_benchmarked_mode)."""
    import inspect
    import re
    from mellow_amd import engine as E, wrapper as W
    assert E.DEFAULT_PRECISION == "f32x3"
    assert inspect.signature(E.Engine.__init__).parameters["precision"].default is None        # None -> MELLOW_PRECISION or the default
    assert inspect.signature(W.MellowWrapper.__init__).parameters["precision"].default is None
    bench = open(os.path.join(ROOT, "bench.py")).read()
    m = re.search(r'add_argument\("--precision", choices=\([^)]*\), default="(\w+)"', bench)
    assert m and m.group(1) == E.DEFAULT_PRECISION
    hdr = open(os.path.join(ROOT, "include", "mellow_hip.h")).read()
    assert "MELLOW_PRECISION_F32X3 (DEFAULT" in hdr and "experimental" not in hdr
    src = open(os.path.join(ROOT, "mellow_amd", "csrc", "engine.cpp")).read()
    assert "return mellow_engine_set_precision(e, MELLOW_PRECISION_F32X3);" in src              # mellow_engine_create


def test_reference_import_name_is_a_drop_in():
    """`from mellow import MellowWrapper` (reference mellow/__init__.py:1, README.md:47) resolves to this build's wrapper."""
    import importlib
    sys.modules.pop("mellow", None)
    sys.path.insert(0, ROOT)
    try:
        m = importlib.import_module("mellow")
        from mellow_amd.wrapper import MellowWrapper as W
        assert m.MellowWrapper is W
        assert importlib.import_module("mellow.wrapper").MellowWrapper is W
        assert os.path.dirname(m.__file__) == os.path.join(ROOT, "mellow")
    finally:
        sys.path.remove(ROOT)


RESAMPLE_CASES = ((48000, 12345), (22050, 7777), (16000, 4000), (22050, 1), (8000, 333))


def _oracle_cases():
    """(name, sample rate, float32 rows) the resampler checks run on: the reference's own fixture clips (resource/1.wav,
    2.wav: 403,604 and 445,940 samples @ 44.1 kHz, the first 60,000 and the last 45,000 samples of each -- both edges of the
    zero extension) and seeded noise at 48 / 22.05 / 16 / 8 kHz with odd lengths."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "example.npz"))
    out = []
    for k in ("pcm1", "pcm2"):
        w = g[k].astype(np.float32) / 32768.0
        out.append((k + "_head", 44100, w[None, :60000]))
        out.append((k + "_tail", 44100, w[None, -45000:]))
    rng = np.random.default_rng(9)
    for sr, n in RESAMPLE_CASES:
        out.append((f"noise_{sr}_{n}", sr, (rng.standard_normal((2, n)) * 0.3).astype(np.float32)))
    return out


def test_host_resampler_against_the_independent_fp64_oracle():
    """f1 / A0's filter step (reference wrapper.py:144-148, torchaudio.transforms.Resample): `mellow_amd.audio.resample` (fp32
    polyphase bank through conv1d) against oracle/resample_oracle.py (fp64, one windowed-sinc sum per output sample, nothing
    shared with the product).  Tolerance = the fp64-derived bound of the oracle -- (taps + 2) * 2^-24 * sum|x||h| per sample,
    what ANY fp32 evaluation of the same sums can differ by -- and, tighter, 1e-6 of the clip's peak (measured: 2-3e-7)."""
    from oracle import resample_oracle as R
    assert "mellow_amd" not in open(R.__file__).read().split('"""', 2)[2]          # the oracle's code imports nothing of the product
    for name, sr, x in _oracle_cases():
        want, bound = R.resample(x, sr, 32000, return_bound=True)
        got = audio.resample(torch.from_numpy(x), sr, 32000).numpy().astype(np.float64)
        assert got.shape == want.shape == (x.shape[0], R.output_length(x.shape[1], sr, 32000)), name
        d = np.abs(got - want)
        assert (d <= R.fp32_tolerance(bound, sr, 32000)).all(), (name, float(d.max()))
        assert d.max() <= 1e-6 * max(1.0, float(np.abs(want).max())), (name, float(d.max()))


def test_resample_oracle_whole_fixture_lengths_and_geometry():
    """the oracle itself: torchaudio's geometry for the reference's fixtures (441 -> 320 after the gcd, width 9, 459 taps; output
    lengths 292,865 and 323,585), the tile of the whole resource/1.wav the reference feeds the model, and closed-form checks
    the product never sees (DC gain, a 1 kHz tone) so that product and oracle cannot be wrong together unnoticed."""
    from oracle import resample_oracle as R
    assert R.geometry(44100, 32000) == (441, 320, 316.8, 9, 459)
    assert R.output_length(403604, 44100, 32000) == 292865 and R.output_length(445940, 44100, 32000) == 323585
    assert R.geometry(16000, 32000)[:2] == (1, 2) and R.geometry(48000, 32000)[:2] == (3, 2)
    n = 22050
    dc = R.resample(np.full((1, n), 0.5), 44100, 32000)[0]
    assert np.abs(dc[200:-200] - 0.5).max() < 1e-3
    t = np.arange(n) / 44100.0
    y = R.resample(np.sin(2 * np.pi * 1000.0 * t)[None], 44100, 32000)[0]
    ref = np.sin(2 * np.pi * 1000.0 * np.arange(len(y)) / 32000.0)
    assert np.abs(y[500:-500] - ref[500:-500]).max() < 1e-3
    g = np.load(os.path.join(ROOT, "tests", "golden", "example.npz"))
    w1 = g["pcm1"].astype(np.float32) / 32768.0
    full = R.resample(w1[None], 44100, 32000)[0].astype(np.float32)                # what the shimmed reference resampled
    tiled = np.concatenate([full, full])[:320000]
    assert np.array_equal(tiled[::61], g["audio1_sub"])                             # the golden's audio1 IS the oracle's output, tiled


def test_resampler_closed_form_properties():
    """PARITY UNPINNED against torchaudio (absent offline).  What any sinc-Hann resampler with torchaudio's defaults must
    satisfy: output length ceil(new*n/orig) (the reference's fixtures: 403,604 @ 44.1 kHz -> 292,865), DC gain 1, an in-band
    tone keeps amplitude/frequency, a tone above the new Nyquist is rejected."""
    for sr, n, want in ((44100, 403604, 292865), (44100, 441, 320), (48000, 3, 2), (16000, 5, 10), (22050, 1, 2)):
        assert audio.resample(torch.zeros(1, n), sr, 32000).shape == (1, want), (sr, n)
    sr, n = 44100, 44100
    dc = audio.resample(torch.full((1, n), 0.5), sr, 32000)[0]
    assert float((dc[200:-200] - 0.5).abs().max()) < 1e-3      # a width-6 windowed sinc has ~5e-4 DC ripple
    t = np.arange(n) / sr
    # the width-6 window gives a wide transition band: 1 kHz passes to 2e-4, 12 kHz loses 1.5 %, the cut-off (0.99 x 16 kHz)
    # is the half-amplitude point, 20 kHz (aliasing to 12 kHz if it leaked) is down to 0.7 %
    for f, tol in ((1000.0, 1e-3), (12000.0, 3e-2)):
        y = audio.resample(torch.from_numpy(np.sin(2 * np.pi * f * t).astype(np.float32))[None], sr, 32000)[0].numpy()
        ref = np.sin(2 * np.pi * f * np.arange(len(y)) / 32000.0)
        assert np.abs(y[500:-500] - ref[500:-500]).max() < tol, f
    y = audio.resample(torch.from_numpy(np.sin(2 * np.pi * 15900.0 * t).astype(np.float32))[None], sr, 32000)[0].numpy()
    assert 0.35 < np.abs(y[500:-500]).max() < 0.65
    y = audio.resample(torch.from_numpy(np.sin(2 * np.pi * 20000.0 * t).astype(np.float32))[None], sr, 32000)[0].numpy()
    assert np.abs(y[500:-500]).max() < 1e-2


def test_non_finite_samples_are_sanitised_and_formats_reported(tmp_path):
    from scipy.io import wavfile
    x = np.zeros(2000, dtype=np.float32)
    x[10], x[20], x[30] = np.nan, np.inf, -np.inf
    p = str(tmp_path / "f32.wav")
    wavfile.write(p, 32000, x)
    with pytest.warns(UserWarning, match="non-finite"):
        w, sr = audio.load_wav(p)
    assert sr == 32000 and torch.isfinite(w).all() and float(w[0, 20]) == 1.0 and float(w[0, 30]) == -1.0 and float(w[0, 10]) == 0.0
    bad = tmp_path / "clip.mp3"
    bad.write_bytes(b"ID3\x03\x00\x00\x00\x00\x00\x00not really audio")
    with pytest.raises(ValueError, match="cannot decode audio.*PCM / float WAV"):
        audio.load_wav(str(bad))


def test_examples_signature_ignores_mtime_and_never_stats_the_prompt(tmp_path, monkeypatch):
    """Data-parallel agreement (dist.examples_signature): two nodes holding COPIES of the same clips (same name and bytes, other
    mtime) must agree; same-named files with other bytes must not; and the prompt slot is text only -- a prompt that happens to
    name a file in one rank's working directory is the same prompt on every rank."""
    import time
    from mellow_amd import dist as mdist
    a, b = tmp_path / "n0", tmp_path / "n1"
    a.mkdir(); b.mkdir()
    payload = (np.arange(200_000, dtype=np.int16)).tobytes()
    for d in (a, b):
        (d / "clip.wav").write_bytes(payload)
    os.utime(b / "clip.wav", (time.time() - 86400, time.time() - 86400))          # the copy on "node 1" is a day older
    sig = {}
    for d in (a, b):
        monkeypatch.chdir(d)
        sig[d] = mdist.examples_signature([["clip.wav", "clip.wav", "what is this?"]])
    assert sig[a] == sig[b]
    (b / "clip.wav").write_bytes(payload[:-2] + b"\x01\x02")                      # same name and size, other tail bytes
    monkeypatch.chdir(b)
    assert mdist.examples_signature([["clip.wav", "clip.wav", "what is this?"]]) != sig[a]
    # a prompt equal to an existing file's name: hashed as text on both ranks, whether or not the file exists there
    monkeypatch.chdir(a)
    (a / "notes.txt").write_bytes(b"x" * 10)
    s_with = mdist.examples_signature([[np.zeros(4, np.float32), np.zeros(4, np.float32), "notes.txt"]])
    monkeypatch.chdir(tmp_path)
    assert mdist.examples_signature([[np.zeros(4, np.float32), np.zeros(4, np.float32), "notes.txt"]]) == s_with
