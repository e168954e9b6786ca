"""TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg) -- never imported by the product.

Independent fp64 oracle for the resampling step of reference A0 (`/root/reference/mellow/wrapper.py:144-148`:
`torchaudio.load` + `T.Resample(sample_rate, resample_rate)`).  torchaudio (pinned 2.0.1 by the reference's
requirements.txt) is NOT installed in this image, so its published algorithm is restated here
as a TIME-DOMAIN definition, one output sample at a time:

    g = gcd(orig_freq, new_freq);  orig, new = orig_freq / g, new_freq / g
    base  = min(orig, new) * 0.99                       (rolloff)
    width = ceil(6 * orig / base)                       (lowpass_filter_width = 6)
    output sample m  (0 <= m < ceil(new * n / orig)),  frame q = m // new, phase j = m % new:
        y[m] = sum over d = -width .. width + orig - 1 of   x[q * orig + d] * h((d / orig - j / new) * base)
        h(t) = sinc(pi * tc) * cos(pi * tc / 12)**2 * (base / orig),   tc = clamp(t, -6, +6)      (x = 0 outside [0, n))

This file deliberately shares nothing with the product's resamplers (`mellow_amd/audio.py`: a polyphase filter bank run
through conv1d in fp32; `mellow_resample`: the HIP twin): no kernel bank is built, no convolution routine is called, nothing
is imported from `mellow_amd`, and every product and sum is fp64.  Parity status: PINNED TO THE PUBLISHED ALGORITHM, NOT TO
torchaudio's BINARY (absent); the golden "example" run of tests/golden/make_golden.py uses THIS file as the reference's
`torchaudio.transforms.Resample` (tests/golden/ref_shims/torchaudio/transforms.py).

`resample(x, orig_freq, new_freq, return_bound=True)` also returns, per output sample, B[m] = sum |x| * |h|: an fp32
implementation that rounds its taps to fp32 and accumulates T = 2 * width + orig terms in fp32 in ANY order differs from the
fp64 value by at most (T + 2) * 2**-24 * B[m] (+ the final rounding of the result) -- the tolerance the tests use.
"""
from __future__ import annotations

import math

import numpy as np

LOWPASS_FILTER_WIDTH = 6
ROLLOFF = 0.99


def geometry(orig_freq: int, new_freq: int):
    """-> (orig, new, base, width, taps) of the published algorithm."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * ROLLOFF
    width = int(math.ceil(LOWPASS_FILTER_WIDTH * orig / base))
    return orig, new, base, width, 2 * width + orig


def _h(t: np.ndarray, base: float, orig: int) -> np.ndarray:
    tc = np.clip(t, -float(LOWPASS_FILTER_WIDTH), float(LOWPASS_FILTER_WIDTH))
    window = np.cos(tc * (math.pi / LOWPASS_FILTER_WIDTH / 2.0)) ** 2
    a = tc * math.pi
    with np.errstate(invalid="ignore", divide="ignore"):
        s = np.where(a == 0.0, 1.0, np.sin(a) / np.where(a == 0.0, 1.0, a))
    return s * window * (base / orig)


def output_length(n: int, orig_freq: int, new_freq: int) -> int:
    orig, new, *_ = geometry(orig_freq, new_freq)
    return int(math.ceil(new * n / orig))


def resample(x, orig_freq: int, new_freq: int, return_bound: bool = False, chunk: int = 4096):
    """x: (..., n) real -> (..., ceil(new * n / orig)) float64 (and the per-sample bound B when asked)."""
    x = np.asarray(x, dtype=np.float64)
    if int(orig_freq) == int(new_freq):
        return (x, np.abs(x)) if return_bound else x
    orig, new, base, width, taps = geometry(orig_freq, new_freq)
    lead = x.shape[:-1]
    n = x.shape[-1]
    rows = x.reshape(-1, n)
    m_total = int(math.ceil(new * n / orig))
    out = np.empty((rows.shape[0], m_total), dtype=np.float64)
    bound = np.empty_like(out) if return_bound else None
    d = np.arange(-width, width + orig, dtype=np.int64)                       # tap offsets inside a frame, (taps,)
    for m0 in range(0, m_total, chunk):
        m = np.arange(m0, min(m_total, m0 + chunk), dtype=np.int64)
        q, j = m // new, m % new
        t = (d[None, :].astype(np.float64) / orig - j[:, None].astype(np.float64) / new) * base      # (M, taps)
        hv = _h(t, base, orig)
        idx = q[:, None] * orig + d[None, :]
        ok = (idx >= 0) & (idx < n)
        idx = np.where(ok, idx, 0)
        for r in range(rows.shape[0]):
            xv = np.where(ok, rows[r][idx], 0.0)
            out[r, m0:m0 + len(m)] = np.sum(xv * hv, axis=1)
            if return_bound:
                bound[r, m0:m0 + len(m)] = np.sum(np.abs(xv) * np.abs(hv), axis=1)
    out = out.reshape(lead + (m_total,))
    if return_bound:
        return out, bound.reshape(lead + (m_total,))
    return out


def fp32_tolerance(bound: np.ndarray, orig_freq: int, new_freq: int) -> np.ndarray:
    """per-sample |fp32 implementation - fp64 oracle| bound (module docstring): taps rounded to fp32 (1 ulp/2 each), products
    and T-term accumulation in fp32 in any order, result rounded to fp32."""
    *_, taps = geometry(orig_freq, new_freq)
    return (taps + 2) * 2.0 ** -24 * np.asarray(bound) + 2.0 ** -126
