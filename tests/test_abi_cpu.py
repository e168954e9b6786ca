"""CPU: the C-ABI library loads, exports every symbol declared in include/mellow_hip.h, and its host-only
helpers (window permutation, weight fragment packing) agree with the reference's torch ops.
No compute call is made: there is no GPU here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from mellow_amd import engine as E
from oracle import mellow_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(E.LIB_PATH):
        from mellow_amd.csrc import build
        build.build()
    return E.load_library()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "mellow_hip.h")).read()
    declared = set(re.findall(r"\b(mellow_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"mellow_config", "mellow_engine"}
    assert declared == set(E.EXPORTED_SYMBOLS), declared ^ set(E.EXPORTED_SYMBOLS)
    raw = ctypes.CDLL(E.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert lib.mellow_abi_version() == E.ABI_VERSION


def test_required_keys_are_the_reference_state_dict_keys(lib):
    from mellow_amd import spec
    layout = spec.state_dict_layout()
    n = lib.mellow_engine_num_required()
    keys = [lib.mellow_engine_required_key(i).decode() for i in range(n)]
    assert len(set(keys)) == n
    assert set(keys) <= set(layout)
    unused = set(layout) - set(keys)
    assert unused == set(spec.UNUSED_KEYS) | {spec.LM + "lm_head.weight"}, unused


def test_engine_create_fails_loudly_without_gpu(lib):
    if lib.mellow_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(E.EngineError, match="no HIP device"):
        E.Engine()


@pytest.mark.parametrize("R,shift", [(64, 0), (64, 4), (32, 4), (16, 0), (16, 4), (8, 0)])
def test_window_map_is_roll_plus_partition(lib, R, shift):
    """row m of the window-ordered batch must be token map[m] (reference htsat.py:427-436)."""
    m = E.host_window_map(R, shift)
    tok = torch.arange(R * R, dtype=torch.float32).view(1, R, R, 1)
    x = torch.roll(tok, shifts=(-shift, -shift), dims=(1, 2)) if shift else tok
    ws = min(R, 8)
    ref = O.window_partition(x, ws).reshape(-1).to(torch.int64).numpy()
    assert np.array_equal(m, ref)
    # the same map scatters back: window_reverse + roll(+shift) is the inverse permutation
    y = torch.zeros(R * R)
    y[torch.from_numpy(m).long()] = torch.arange(R * R, dtype=torch.float32)
    back = O.window_reverse(torch.arange(R * R, dtype=torch.float32).view(-1, ws, ws, 1), ws, R, R)
    back = torch.roll(back, shifts=(shift, shift), dims=(1, 2)) if shift else back
    assert torch.equal(y, back.reshape(-1))


def test_pack_weight_fragment_order(lib):
    rng = np.random.default_rng(0)
    N, K = 70, 52
    w = rng.standard_normal((N, K)).astype(np.float32)
    p = E.host_pack_weight(w, npad=128)
    assert p.shape == (4, 8, 64, 4)
    for nt, k8, lane, j in [(0, 0, 0, 0), (1, 3, 37, 2), (2, 6, 5, 3), (2, 6, 63, 3), (0, 7, 40, 1), (3, 0, 0, 0)]:
        n, k = nt * 32 + (lane & 31), k8 * 8 + 4 * (lane >> 5) + j
        want = w[n, k] if (n < N and k < K) else 0.0
        assert p[nt, k8, lane, j] == want
    # mfma k-pairing: the 4 MFMAs of one float4 cover k0+j and k0+4+j -> all 8 k of the tile exactly once
    ks = sorted(k8 * 8 + 4 * h + j for k8 in range(1) for h in range(2) for j in range(4))
    assert ks == list(range(8))


def test_rope_tables_helper_against_the_hf_computation(lib):
    """`mellow_host_rope_tables` is what the engine uses when the binding does not load "mellow.rope_cos/sin"
    (include/mellow_hip.h): same fp32 inv_freq / angle as transformers' LlamaRotaryEmbedding, cos / sin correctly rounded.
    torch's vectorised fp32 cos / sin are allowed to differ from that by one unit in the last place, nothing more."""
    P, D, theta = 2048, 64, 100000.0
    c = np.empty((P, D // 2), dtype=np.float32)
    s = np.empty((P, D // 2), dtype=np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    assert lib.mellow_host_rope_tables(theta, D, P, c.ctypes.data_as(fp), s.ctypes.data_as(fp)) == 0
    hc, hs = E.hf_rope_tables(P, D, theta)
    # the angle path is bit-identical: position 1 holds cos / sin of inv_freq itself, and exact identities hold at position 0
    assert np.array_equal(c[0], np.ones(D // 2, dtype=np.float32)) and np.array_equal(s[0], np.zeros(D // 2, dtype=np.float32))
    for got, ref in ((c, hc), (s, hs)):
        ulp = np.abs(got.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))
        assert ulp.max() <= 1, ulp.max()
        assert (ulp != 0).mean() < 0.10
    # and against float64: the helper is the correctly rounded value
    inv = (1.0 / (theta ** (torch.arange(0, D, 2, dtype=torch.int64).float() / D))).numpy()
    ang = (np.arange(P, dtype=np.float32)[:, None] * inv[None, :]).astype(np.float32).astype(np.float64)
    assert np.array_equal(c, np.cos(ang).astype(np.float32)) and np.array_equal(s, np.sin(ang).astype(np.float32))
    assert lib.mellow_host_rope_tables(theta, 63, P, c.ctypes.data_as(fp), s.ctypes.data_as(fp)) != 0
