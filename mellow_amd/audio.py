"""Host audio ingest: the part of `MellowWrapper.load_audio_into_tensor` (reference wrapper.py:141-168) that
the reference delegates to torchaudio (`torchaudio.load`, `torchaudio.transforms.Resample`), restated on
numpy/torch-CPU because torchaudio is not a dependency of this package.

PARITY UNPINNED for the resampler: torchaudio 2.0.1 is not available offline, so the sinc-hann polyphase
kernel below restates its published algorithm (lowpass_filter_width 6, rolloff 0.99,
gcd-reduced rates, output length ceil(new*n/orig)); at the 32 kHz rate the model is built for it is a no-op.
"""
from __future__ import annotations

import math
import random
import wave
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F


def _decode_file(path: str) -> Tuple[np.ndarray, int]:
    """-> (float32 (channels, n), sample_rate).  Decoders are tried in the order torchaudio (what the reference uses,
    wrapper.py:144) -> soundfile -> scipy.io.wavfile -> stdlib wave; the first two are optional dependencies."""
    errors = []
    try:
        import torchaudio  # optional: not a dependency of this package
        wav, sr = torchaudio.load(path)
        return wav.numpy().astype(np.float32, copy=False), int(sr)
    except ImportError:
        pass
    except Exception as e:  # pragma: no cover - depends on the installed backend
        errors.append(f"torchaudio: {e}")
    try:
        import soundfile  # optional: FLAC / OGG / WAV-extensible
        data, sr = soundfile.read(path, dtype="float32", always_2d=True)
        return np.ascontiguousarray(data.T), int(sr)
    except ImportError:
        pass
    except Exception as e:  # pragma: no cover
        errors.append(f"soundfile: {e}")
    try:
        from scipy.io import wavfile
        sr, data = wavfile.read(path)
        if data.ndim == 1:
            data = data[:, None]
        if data.dtype == np.int16:
            x = data.astype(np.float32) / 32768.0
        elif data.dtype == np.int32:
            x = data.astype(np.float32) / 2147483648.0
        elif data.dtype == np.uint8:
            x = (data.astype(np.float32) - 128.0) / 128.0
        else:
            x = data.astype(np.float32)
        return np.ascontiguousarray(x.T), int(sr)
    except ImportError:  # pragma: no cover - scipy is present in the image
        pass
    except Exception as e:
        errors.append(f"scipy.io.wavfile: {e}")
    try:
        with wave.open(path, "rb") as w:
            sr, ch, sw, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
            if sw != 2:
                raise ValueError("only 16-bit PCM is supported by the stdlib decoder")
            data = np.frombuffer(w.readframes(n), dtype="<i2").reshape(-1, ch)
        return (data.astype(np.float32) / 32768.0).T.copy(), sr
    except Exception as e:
        errors.append(f"wave: {e}")
    raise ValueError(f"{path}: cannot decode audio.  Without torchaudio or soundfile installed only PCM / float WAV files "
                     f"are supported (FLAC, MP3, OGG need one of them).  Decoder errors: " + "; ".join(errors))


def load_wav(path: str) -> Tuple[torch.Tensor, int]:
    """-> (float32 tensor (channels, n) in [-1, 1), sample_rate), like torchaudio.load(normalize=True).
    Non-finite samples (float WAVs can hold NaN / Inf) are replaced by 0 / +-1 with a warning: one NaN would poison every
    logit of its example (the reference would emit garbage text for it; an engine must not fault on it)."""
    x, sr = _decode_file(path)
    if not np.isfinite(x).all():
        import warnings
        warnings.warn(f"{path}: non-finite samples replaced (NaN -> 0, +-Inf -> +-1)")
        x = np.nan_to_num(x, nan=0.0, posinf=1.0, neginf=-1.0)
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)), int(sr)


def _sinc_resample_kernel(orig: int, new: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """Provenance: a close restatement, step for step, of the sinc-kernel routine of torchaudio (third party, BSD 2-Clause:
    Copyright (c) 2017 Facebook Inc. (Soumith Chintala), licence text in /NOTICE; pinned by the reference at 2.0.1 -- NOT a
    file of the reference repository), kept that close on purpose: the resampled samples are the encoder's input, so any other
    windowing would be a different model input.  Checked against an independent fp64 time-domain evaluation of the same
    published algorithm (oracle/resample_oracle.py) in tests/test_host_cpu.py and, for the device twin, tests/test_gpu_parity.py."""
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = t * base_freq
    t = t.clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base_freq / orig
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t)
    kernels = kernels * window * scale
    return kernels.to(torch.float32), width


def resample(waveform: torch.Tensor, orig_freq: int, new_freq: int) -> torch.Tensor:
    """sinc_interp_hann resampling of (channels, n) float32 (torchaudio.transforms.Resample defaults)."""
    if orig_freq == new_freq:
        return waveform
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    kernel, width = _sinc_resample_kernel(orig, new)
    shape = waveform.shape
    x = waveform.reshape(-1, shape[-1])
    length = x.shape[1]
    x = F.pad(x, (width, width + orig))
    y = F.conv1d(x[:, None], kernel, stride=orig)
    y = y.transpose(1, 2).reshape(x.shape[0], -1)
    target = int(math.ceil(new * length / orig))
    return y[..., :target].reshape(shape[:-1] + (target,))


def fit_duration(x: torch.Tensor, n_target: int, start_index: Optional[int] = None) -> torch.Tensor:
    """Tile (n <= target) or random-crop (n > target) a flat waveform, reference wrapper.py:152-167.
    `start_index` injects the crop offset (the reference draws it from the unseeded `random` module)."""
    n = x.shape[0]
    if n_target >= n:
        repeat_factor = int(np.ceil(n_target / n))
        x = x.repeat(repeat_factor)
        return x[0:n_target]
    if start_index is None:
        start_index = random.randrange(n - n_target)
    return x[start_index:start_index + n_target]


def load_audio_into_tensor(audio, audio_duration: int, sampling_rate: int, resample_audio: bool = True,
                           start_index: Optional[int] = None) -> torch.Tensor:
    """reference wrapper.py:141-168.  `audio` is a wav path, or (extension) a 1-D/2-D float array at
    `sampling_rate`.  Multi-channel audio is flattened channel after channel, not mixed (wrapper.py:149)."""
    if isinstance(audio, (str, bytes)) or hasattr(audio, "__fspath__"):
        wav, sr = load_wav(str(audio))
        if resample_audio and sr != sampling_rate:
            wav = resample(wav, sr, sampling_rate)
    else:
        wav = torch.as_tensor(np.asarray(audio), dtype=torch.float32)
    wav = wav.reshape(-1)
    return fit_duration(wav, audio_duration * sampling_rate, start_index).to(torch.float32)
