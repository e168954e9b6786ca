"""GPU (-m gpu): INTEGRATION.md sections 1-2 executed verbatim with nothing but `ctypes` (no mellow_amd.engine): the binding a
maintainer of the reference would write against include/mellow_hip.h.  Checked against the reference's own 300-step loop
(tests/golden/late.npz), with the optional "mellow.rope_cos/sin" tensors and without them (the in-library tables)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from mellow_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "mellow_amd", "lib", "libmellow_hip.so")


class Cfg(C.Structure):                      # mellow_config_t
    _fields_ = [(n, C.c_int32) for n in ("abi_version", "vocab_size", "hidden_size", "intermediate_size",
                "num_layers", "num_heads", "num_kv_heads", "head_dim")] + \
               [("rms_norm_eps", C.c_float), ("rope_theta", C.c_float)] + \
               [(n, C.c_int32) for n in ("max_positions", "text_len", "prefix_len", "sep_token_id")]


@pytest.mark.parametrize("host_rope", [True, False], ids=["rope_from_torch", "rope_in_library"])
@pytest.mark.parametrize("precision", [0, 2, None], ids=["f32", "f32x3", "library_default"])
def test_raw_ctypes_binding_reproduces_the_reference_loop(golden_dir, host_rope, precision, monkeypatch):
    lib = C.CDLL(LIB)
    lib.mellow_last_error.restype = C.c_char_p

    def chk(rc):
        if rc:
            raise RuntimeError(lib.mellow_last_error().decode())

    # ---- INTEGRATION.md section 1 --------------------------------------------------------------------------------------
    device = 0
    cfg = Cfg(2, 49152, 576, 1536, 30, 9, 3, 64, 1e-5, 100000.0, 2048, 129, 389, 0)
    h = C.c_void_p()
    # (variables that switched kernels or numeric forms up to round 5: the round-6 library must ignore them)
    for k, v in (("MELLOW_SPLITK", "0"), ("MELLOW_DECODE_X3", "0"), ("MELLOW_PREFILL_SPLIT", "1"), ("MELLOW_DECODE_FUSE", "0"), ("MELLOW_F32X3_TERMS", "9")):
        monkeypatch.setenv(k, v)
    chk(lib.mellow_engine_create(C.byref(cfg), device, C.byref(h)))
    if precision is not None:                                               # no call: the library's default mode (f32x3)
        chk(lib.mellow_engine_set_precision(h, precision))                 # section 6
    state = synth.make_state_dict(0)                                        # stands for torch.load(self.model_path)
    for key, t in state.items():
        t = t.contiguous()
        dt = {torch.float32: 0, torch.int32: 1, torch.int64: 2}[t.dtype]
        shape = (C.c_int64 * max(1, t.dim()))(*t.shape)
        chk(lib.mellow_engine_load_tensor(h, key.encode(), C.c_void_p(t.data_ptr()), shape, t.dim(), dt))
    if host_rope:
        inv_freq = 1.0 / (100000.0 ** (torch.arange(0, 64, 2, dtype=torch.int64).float() / 64))
        freqs = torch.outer(torch.arange(2048, dtype=torch.float32), inv_freq)
        for name, t in (("mellow.rope_cos", freqs.cos().contiguous()), ("mellow.rope_sin", freqs.sin().contiguous())):
            chk(lib.mellow_engine_load_tensor(h, name.encode(), C.c_void_p(t.data_ptr()), (C.c_int64 * 2)(*t.shape), 2, 0))
    chk(lib.mellow_engine_finalize(h))

    # ---- INTEGRATION.md section 2 --------------------------------------------------------------------------------------
    g = np.load(os.path.join(golden_dir, "late.npz"))
    max_len = int(g["steps"])
    a1, a2, idn = synth.make_batch(2)
    dev = torch.device(f"cuda:{device}")
    audio1 = torch.from_numpy(a1).to(dev).float().contiguous()
    audio2 = torch.from_numpy(a2).to(dev).float().contiguous()
    ids = torch.from_numpy(idn).to(dev).to(torch.int32).contiguous()
    torch.cuda.current_stream(dev).synchronize()
    B = ids.shape[0]
    out = torch.empty((B, max_len), dtype=torch.int32, device=ids.device)
    lens = (C.c_int32 * B)(); steps = C.c_int32(); first_ms = C.c_float()
    stop = -1                                                                # the synthetic tokenizer-free run: never stops
    chk(lib.mellow_generate(h, C.c_void_p(audio1.data_ptr()), C.c_void_p(audio2.data_ptr()), C.c_int64(audio1.shape[1]),
                            C.c_void_p(ids.data_ptr()), B, max_len, C.c_float(0.8), C.c_float(1.0), stop, 0,
                            C.c_void_p(out.data_ptr()), lens, C.byref(steps), C.byref(first_ms)))
    tokens = out[:, :steps.value].cpu().numpy()
    # the engine owns its stream configuration: a raw C-ABI consumer gets the two-stream prefill in
    # the f32x3 modes because the engine measured that its side stream overlaps the main one -- not because of an import
    lib.mellow_prefill_parts.argtypes = [C.c_void_p]
    parts = lib.mellow_prefill_parts(h)
    if precision == 0:
        assert parts == 1, parts
    elif parts != 2:        # a shared GPU may leave no free hardware queue while the probe runs: legal (one chain), but say so
        import warnings
        assert parts == 1, parts
        warnings.warn("the engine's stream probe found no overlapping side stream on this box: the prefill ran as one chain")
    # the library's configuration is explicit: with no mellow_engine_set_option call every option is at
    # its default -- whatever MELLOW_* variables the process environment holds (the library reads none of them)
    lib.mellow_engine_describe.restype = C.c_int64
    lib.mellow_engine_describe.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    n = lib.mellow_engine_describe(h, None, 0)
    buf = C.create_string_buffer(n)
    assert lib.mellow_engine_describe(h, buf, n) == n
    import json
    d = json.loads(buf.value.decode())
    assert d["reads_environment"] is False and d["non_default"] == [] and d["finalized"] is True, d
    assert d["precision"] == {None: "f32x3", 0: "f32", 2: "f32x3"}[precision] and d["abi"] == [2, lib.mellow_abi_minor()], d
    assert all(o["value"] == o["default"] for o in d["options"].values()), d
    lib.mellow_engine_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    assert lib.mellow_engine_set_option(h, b"splitk", b"0") != 0 and b"before mellow_engine_finalize" in lib.mellow_last_error()
    lib.mellow_engine_destroy.argtypes = [C.c_void_p]
    lib.mellow_engine_destroy(h)
    assert steps.value == max_len and first_ms.value > 0
    bad = np.argwhere(tokens != g["tokens"])
    assert bad.size == 0, f"first divergence from the reference at (row, step) {bad[0].tolist()}"
