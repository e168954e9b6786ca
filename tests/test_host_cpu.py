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
    drew from `random` under the stored seed.  The resampling step inside both sides is this build's restatement of torchaudio's
    (PARITY UNPINNED there: torchaudio is not installed); decode, flatten, tile, crop and the draw are pinned to the reference."""
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
    for flag in ("--gpus", "--steps", "--warmup", "--precision", "--inflight", "--no-cpu-baseline"):
        assert flag in r.stdout


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
