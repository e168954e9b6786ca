"""Seeded synthetic checkpoint + synthetic inputs with the reference's exact `state_dict` layout.

Real `v0.ckpt` / `v0_s.ckpt` cannot be fetched offline (reference wrapper.py:41 downloads them from the
HF hub), so parity tests and `bench.py` run on a deterministic synthetic checkpoint whose key names,
shapes and dtypes are those of the real one.  The same dict is loaded (strict) into the
imported reference when goldens are generated (tests/golden/make_golden.py) and into the HIP engine.

Every tensor is drawn from its own numpy PCG64 stream keyed by (seed, crc32(key)), so any subset can be
regenerated independently and the values do not depend on generation order.

Constants that are *frozen parameters* in the reference are the true ones, not noise:
  * conv_real / conv_imag = hann-windowed DFT basis   (torchlibrosa STFT)
  * melW = Slaney mel filterbank 50..14000 Hz          (librosa.filters.mel)
  * relative_position_index / attn_mask               (reference htsat.py:277-291, 389-412)
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np
import torch

from . import spec
from .spec import LMConfig


def _rng(seed: int, key: str) -> np.random.Generator:
    return np.random.default_rng([int(seed) & 0x7FFFFFFF, zlib.crc32(key.encode())])


# ---- frozen constants ------------------------------------------------------------------------------
def dft_conv_weights():
    """hann(periodic) * DFT basis as conv1d weights (513,1,1024): real and imaginary parts."""
    n = np.arange(spec.WINDOW_SIZE, dtype=np.float64)
    window = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / spec.WINDOW_SIZE)  # periodic hann
    k = np.arange(spec.N_FREQ, dtype=np.float64)
    ang = -2.0 * np.pi * np.outer(k, n) / spec.WINDOW_SIZE           # W[n,k] = exp(-2*pi*i*n*k/N)
    real = (np.cos(ang) * window[None, :]).astype(np.float32)
    imag = (np.sin(ang) * window[None, :]).astype(np.float32)
    return real[:, None, :], imag[:, None, :]


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def slaney_mel_filterbank():
    """(513, 64) Slaney-normalised triangular mel filterbank, fmin 50, fmax 14000, sr 32000."""
    fftfreqs = np.linspace(0, spec.SAMPLE_RATE / 2, spec.N_FREQ)
    mel_pts = np.linspace(_hz_to_mel(spec.FMIN), _hz_to_mel(spec.FMAX), spec.MEL_BINS + 2)
    mel_f = _mel_to_hz(mel_pts)
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((spec.MEL_BINS, spec.N_FREQ))
    for i in range(spec.MEL_BINS):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:spec.MEL_BINS + 2] - mel_f[:spec.MEL_BINS])
    w *= enorm[:, None]
    return w.T.astype(np.float32).copy()


def relative_position_index():
    """(64,64) int64 index into the (225,nH) bias table — the arithmetic of reference htsat.py:281-291."""
    ws = spec.WINDOW
    ch, cw = np.meshgrid(np.arange(ws), np.arange(ws), indexing="ij")
    coords = np.stack([ch.reshape(-1), cw.reshape(-1)])          # 2, 64
    rel = coords[:, :, None] - coords[:, None, :]                # 2, 64, 64
    rel = rel.transpose(1, 2, 0).copy()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1).astype(np.int64)


def shifted_window_mask(res: int):
    """(nW,64,64) 0 / -100 mask of a shifted block at resolution `res` (reference htsat.py:389-410)."""
    ws, sh = spec.WINDOW, spec.WINDOW // 2
    img = np.zeros((res, res), dtype=np.float32)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -sh), slice(-sh, None)):
        for wsl in (slice(0, -ws), slice(-ws, -sh), slice(-sh, None)):
            img[hs, wsl] = cnt
            cnt += 1
    mw = img.reshape(res // ws, ws, res // ws, ws).transpose(0, 2, 1, 3).reshape(-1, ws * ws)
    diff = mw[:, None, :] - mw[:, :, None]
    return np.where(diff != 0, np.float32(-100.0), np.float32(0.0)).astype(np.float32)


# ---- synthetic state_dict ---------------------------------------------------------------------------
def _decaying_spectrum(g: np.random.Generator, shape, std: float, alpha: float) -> np.ndarray:
    """A matrix of the given shape and element std whose singular values fall off like i^-alpha (a product of two Gaussian
    factors around a decaying diagonal): what trained weight matrices look like, unlike an i.i.d. Gaussian's flat spectrum."""
    n, k = int(shape[0]), int(np.prod(shape[1:]))
    r = min(n, k)
    sv = (1.0 + np.arange(r, dtype=np.float64)) ** (-alpha)
    a = (g.standard_normal((n, r)).astype(np.float32) * sv[None, :].astype(np.float32)) @ g.standard_normal((r, k)).astype(np.float32)
    a *= np.float32(std / max(float(a.std()), 1e-30))
    return a.reshape(shape)


def make_state_dict(seed: int = 0, lm: LMConfig | None = None, structured: bool = False) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic synthetic checkpoint with the real key layout.

    structured=True (a DIFFERENT checkpoint, used only to put a number on the fp8 mode): every learned matrix
    keeps its scale but gets a decaying singular spectrum (sigma_i ~ i^-1) instead of an i.i.d. Gaussian's flat one, and the
    norm scales are exactly 1 -- closer to a trained network, where a quantisation error does not get re-amplified by every
    layer.  The goldens and every parity test use the default (structured=False)."""
    lm = lm or LMConfig()
    layout = spec.state_dict_layout(lm)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    real, imag = dft_conv_weights()
    rpi = relative_position_index()
    embed_key = spec.LM + "model.embed_tokens.weight"

    for key, (shape, dt) in layout.items():
        g = _rng(seed, key)
        leaf = key.split(".")[-1]
        if key.endswith("conv_real.weight"):
            a = real
        elif key.endswith("conv_imag.weight"):
            a = imag
        elif key.endswith("melW"):
            a = slaney_mel_filterbank()
        elif key.endswith("relative_position_index"):
            a = rpi
        elif key.endswith("attn_mask"):
            s = int(key.split("layers.")[1].split(".")[0])
            a = shifted_window_mask(spec.STAGE_RES[s])
        elif key.endswith("num_batches_tracked"):
            a = np.asarray(1000, dtype=np.int64)
        elif key.endswith("bn0.running_mean"):
            a = (-28.0 + 6.0 * g.standard_normal(shape)).astype(np.float32)
        elif key.endswith("bn0.running_var"):
            a = (180.0 * np.exp(0.3 * g.standard_normal(shape))).astype(np.float32)
        elif key.endswith("relative_position_bias_table"):
            a = (0.5 * g.standard_normal(shape)).astype(np.float32)
        elif key == spec.LM + "lm_head.weight":
            a = None  # tied below
        elif key == embed_key:
            a = g.standard_normal(shape).astype(np.float32)           # unit-variance rows, like the LN'd audio rows
        elif key == spec.LM + "model.norm.weight":
            # random signs: with tied embeddings a positive final norm makes "repeat the last token" a
            # fixed point of a random-weight LM (self-logit |E[t]|^2); signs remove that attractor
            sign = np.where(g.random(shape) < 0.5, -1.0, 1.0)
            a = (sign * (1.0 + 0.15 * g.standard_normal(shape))).astype(np.float32)
        elif key.startswith(spec.LM) and (key.endswith("q_proj.weight") or key.endswith("k_proj.weight")):
            fan_in = int(np.prod(shape[1:]))                          # gain 2: peaky, context-dependent attention
            a = (2.0 * g.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)
        elif key.startswith(spec.LM) and (key.endswith("o_proj.weight") or key.endswith("down_proj.weight")):
            fan_in = int(np.prod(shape[1:]))                          # gain 0.5: bounded residual growth over 30 layers
            a = (0.5 * g.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)
        elif leaf == "weight" and len(shape) == 1:
            # LayerNorm / RMSNorm / BatchNorm scale
            a = (1.0 + 0.15 * g.standard_normal(shape)).astype(np.float32)
        elif leaf == "bias":
            a = (0.1 * g.standard_normal(shape)).astype(np.float32)
        elif leaf == "weight":
            fan_in = int(np.prod(shape[1:]))
            a = (g.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)
        else:
            raise KeyError(f"no synthetic rule for {key}")
        if structured and a is not None and a.dtype == np.float32:
            frozen = key.endswith(("conv_real.weight", "conv_imag.weight", "melW", "attn_mask", "running_mean", "running_var"))
            if leaf == "weight" and len(shape) >= 2 and not frozen:
                a = _decaying_spectrum(_rng(seed + 7919, key), shape, float(a.std()), 1.0).astype(np.float32)
            elif leaf == "weight" and len(shape) == 1 and key != spec.LM + "model.norm.weight":
                a = np.ones(shape, dtype=np.float32)
        if a is not None:
            assert tuple(a.shape) == tuple(shape), (key, a.shape, shape)
            sd[key] = torch.from_numpy(np.ascontiguousarray(a)).reshape(shape)
    sd[spec.LM + "lm_head.weight"] = sd[embed_key]  # tied storage, as in the reference
    # keep insertion order of the layout
    return OrderedDict((k, sd[k]) for k in layout.keys())


# ---- synthetic inputs ------------------------------------------------------------------
def make_clip(idx: int, n_samples: int = 320000) -> np.ndarray:
    """One mono f32 clip: U(-0.1,0.1) noise + two sine tones; generator seed 1000+idx."""
    g = np.random.default_rng(1000 + int(idx))
    t = np.arange(n_samples, dtype=np.float64) / spec.SAMPLE_RATE
    f1 = 200.0 + 3000.0 * g.random()
    f2 = 500.0 + 8000.0 * g.random()
    x = g.uniform(-0.1, 0.1, n_samples)
    x += 0.2 * np.sin(2 * np.pi * f1 * t + g.random()) + 0.1 * np.sin(2 * np.pi * f2 * t * (1 + 0.05 * t / t[-1]))
    # slow amplitude envelope so frames differ over time
    x *= 0.6 + 0.4 * np.sin(2 * np.pi * (0.3 + g.random()) * t)
    return x.astype(np.float32)


def make_prompt_ids(idx: int, n_tokens: int = 16, pad_id: int = 17, vocab: int = 49152) -> np.ndarray:
    """16 ids ~ U{17..vocab-1} (seed 2000+idx), right-padded with the '!' id to 129 (wrapper.py:186-190)."""
    g = np.random.default_rng(2000 + int(idx))
    ids = np.full((spec.TEXT_LEN,), pad_id, dtype=np.int64)
    ids[:n_tokens] = g.integers(17, vocab, size=n_tokens)
    return ids


def make_batch(B: int, n_samples: int = 320000, first: int = 0, vocab: int = 49152):
    """(audio1 (B,n), audio2 (B,n), input_ids (B,129)) for examples first..first+B-1."""
    a1 = np.stack([make_clip(2 * (first + i), n_samples) for i in range(B)])
    a2 = np.stack([make_clip(2 * (first + i) + 1, n_samples) for i in range(B)])
    ids = np.stack([make_prompt_ids(first + i, vocab=vocab) for i in range(B)])
    return a1, a2, ids


def make_examples(indices, n_samples: int = 320000, vocab: int = 49152):
    """The same as make_batch for an arbitrary list of example indices (rows in the given order)."""
    parts = [make_batch(1, n_samples=n_samples, first=int(i), vocab=vocab) for i in indices]
    return tuple(np.concatenate([p[k] for p in parts]) for k in range(3))
