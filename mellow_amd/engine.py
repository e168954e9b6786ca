"""ctypes binding of libmellow_hip.so (C ABI in include/mellow_hip.h).

This is the only way Python reaches the hot path.  There is NO CPU fallback: if the shared library is
missing, or no HIP device is visible, constructing an `Engine` raises — the product path never routes
through oracle/ or eager PyTorch.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import spec
from .spec import LMConfig

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libmellow_hip.so")
ABI_VERSION = 2
DEFAULT_PRECISION = "f32x3"      # = MELLOW_PRECISION_F32X3, what mellow_engine_create selects (include/mellow_hip.h)

_F32, _I32, _I64 = 0, 1, 2


class MellowConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("vocab_size", C.c_int32), ("hidden_size", C.c_int32),
        ("intermediate_size", C.c_int32), ("num_layers", C.c_int32), ("num_heads", C.c_int32),
        ("num_kv_heads", C.c_int32), ("head_dim", C.c_int32), ("rms_norm_eps", C.c_float),
        ("rope_theta", C.c_float), ("max_positions", C.c_int32), ("text_len", C.c_int32),
        ("prefix_len", C.c_int32), ("sep_token_id", C.c_int32),
    ]


class EngineError(RuntimeError):
    pass


_lib = None


def load_library(path: Optional[str] = None):
    """dlopen libmellow_hip.so and declare every symbol of include/mellow_hip.h.  Raises if missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("MELLOW_HIP_LIB") or LIB_PATH
    if not os.path.exists(p):
        raise EngineError(
            f"{p} not found: build it with `python mellow_amd/csrc/build.py` (hipcc, gfx950). "
            "The Mellow engine has no CPU fallback.")
    lib = C.CDLL(p)
    vp, ci, cf, i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64
    P = C.POINTER
    sig = {
        "mellow_abi_version": (ci, []),
        "mellow_last_error": (C.c_char_p, []),
        "mellow_device_count": (ci, []),
        "mellow_engine_create": (ci, [P(MellowConfig), ci, P(vp)]),
        "mellow_engine_destroy": (None, [vp]),
        "mellow_engine_fork": (ci, [vp, P(vp)]),
        "mellow_engine_load_tensor": (ci, [vp, C.c_char_p, vp, P(i64), ci, ci]),
        "mellow_engine_finalize": (ci, [vp]),
        "mellow_engine_num_required": (ci, []),
        "mellow_engine_required_key": (C.c_char_p, [ci]),
        "mellow_generate": (ci, [vp, vp, vp, i64, vp, ci, ci, cf, cf, ci, ci, vp, P(C.c_int32), P(C.c_int32), P(cf)]),
        "mellow_logmel": (ci, [vp, vp, ci, i64, ci, vp]),
        "mellow_encode": (ci, [vp, vp, ci, i64, vp]),
        "mellow_prefix": (ci, [vp, vp, vp, i64, vp, ci, vp]),
        "mellow_lm_prefill": (ci, [vp, vp, ci, ci, ci, vp]),
        "mellow_lm_decode_step": (ci, [vp, vp, vp]),
        "mellow_argmax": (ci, [vp, vp, ci, vp]),
        "mellow_embed_tokens": (ci, [vp, vp, ci, vp]),
        "mellow_lm_forward_logits": (ci, [vp, vp, ci, ci, ci, vp]),
        "mellow_resample": (ci, [vp, vp, ci, i64, ci, ci, vp, i64, P(i64)]),
        "mellow_debug_enable_taps": (ci, [vp, ci]),
        "mellow_debug_tap": (ci, [vp, C.c_char_p, vp, i64, P(i64)]),
        "mellow_debug_gemm_fp8": (ci, [vp, vp, ci, ci, vp, ci, vp, ci, vp]),
        "mellow_debug_gemm_f32": (ci, [vp, ci, vp, ci, ci, vp, ci, vp, ci, vp]),
        "mellow_debug_dec_head": (ci, [vp, vp, ci, ci, vp]),
        "mellow_prof_enable": (ci, [vp, ci]),
        "mellow_prof_reset": (ci, [vp]),
        "mellow_prof_num_families": (ci, []),
        "mellow_prof_family_name": (C.c_char_p, [ci]),
        "mellow_prof_get": (ci, [vp, ci, P(i64), P(C.c_double), P(C.c_double), P(C.c_double)]),
        "mellow_last_phase_ms": (ci, [vp, P(cf), P(cf), P(cf)]),
        "mellow_last_steps_enqueued": (ci, [vp]),
        "mellow_last_row_repacks": (ci, [vp]),
        "mellow_stft_is_fft": (ci, [vp]),
        "mellow_prefill_parts": (ci, [vp]),
        "mellow_abi_minor": (ci, []),
        "mellow_engine_set_precision": (ci, [vp, ci]),
        "mellow_engine_set_option": (ci, [vp, C.c_char_p, C.c_char_p]),
        "mellow_engine_describe": (i64, [vp, C.c_char_p, i64]),
        "mellow_set_graph": (ci, [vp, ci]),
        "mellow_host_window_map": (ci, [ci, ci, P(C.c_int32)]),
        "mellow_host_pack_weight": (ci, [P(cf), ci, ci, ci, P(cf), i64]),
        "mellow_host_rope_tables": (ci, [cf, ci, ci, P(cf), P(cf)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.mellow_abi_version() != ABI_VERSION:
        raise EngineError(f"ABI mismatch: library {lib.mellow_abi_version()}, binding {ABI_VERSION}")
    if path is None:
        _lib = lib
    return lib


EXPORTED_SYMBOLS = (
    "mellow_abi_version", "mellow_last_error", "mellow_device_count", "mellow_engine_create",
    "mellow_engine_destroy", "mellow_engine_fork", "mellow_engine_load_tensor", "mellow_engine_finalize",
    "mellow_engine_num_required", "mellow_engine_required_key", "mellow_generate", "mellow_logmel",
    "mellow_encode", "mellow_prefix", "mellow_lm_prefill", "mellow_lm_decode_step", "mellow_argmax", "mellow_embed_tokens", "mellow_lm_forward_logits",
    "mellow_debug_enable_taps", "mellow_debug_tap", "mellow_prof_enable", "mellow_prof_reset",
    "mellow_prof_num_families", "mellow_prof_family_name", "mellow_prof_get", "mellow_last_phase_ms", "mellow_last_steps_enqueued", "mellow_last_row_repacks", "mellow_stft_is_fft", "mellow_prefill_parts", "mellow_abi_minor",
    "mellow_resample", "mellow_engine_set_precision", "mellow_engine_set_option", "mellow_engine_describe", "mellow_debug_gemm_fp8", "mellow_debug_gemm_f32", "mellow_debug_dec_head", "mellow_set_graph", "mellow_host_window_map", "mellow_host_pack_weight", "mellow_host_rope_tables",
)


def hf_rope_tables(max_pos: int, head_dim: int, theta: float):
    """cos/sin [max_pos][head_dim/2] computed exactly the way transformers' LlamaRotaryEmbedding does
    (fp32 inv_freq, fp32 outer product, fp32 cos/sin) so the engine uses bit-identical tables."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    pos = torch.arange(max_pos, dtype=torch.float32)
    freqs = (inv_freq[None, :, None].float() @ pos[None, None, :].float()).transpose(1, 2)[0]
    return freqs.cos().contiguous().numpy(), freqs.sin().contiguous().numpy()


def _ptr(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(t.data_ptr())


class Engine:
    """One engine per device.  Inputs/outputs are torch tensors on that device (plumbing only)."""

    def __init__(self, lm: Optional[LMConfig] = None, device: int = 0, max_positions: Optional[int] = None,
                 precision: Optional[str] = None, options: Optional[Dict[str, int]] = None):
        self.lib = load_library()
        if self.lib.mellow_device_count() <= 0:
            raise EngineError("no HIP device visible: the Mellow engine needs an MI355X (no CPU fallback)")
        self.lm = lm or LMConfig.load()
        self.device = int(device)
        self.tdev = torch.device(f"cuda:{self.device}")
        # positions the KV pages / RoPE tables may reach: the LM's own limit (8192 for SmolLM2-135M) unless the caller asks for
        # less -- the reference's loop is bounded by nothing else (wrapper.py:216, decoder.py:25)
        max_positions = int(max_positions) if max_positions else int(self.lm.max_position_embeddings)
        cfg = MellowConfig(
            abi_version=ABI_VERSION, vocab_size=self.lm.vocab_size, hidden_size=self.lm.hidden_size,
            intermediate_size=self.lm.intermediate_size, num_layers=self.lm.num_hidden_layers,
            num_heads=self.lm.num_attention_heads, num_kv_heads=self.lm.num_key_value_heads,
            head_dim=self.lm.head_dim, rms_norm_eps=self.lm.rms_norm_eps, rope_theta=self.lm.rope_theta,
            max_positions=int(max_positions), text_len=spec.TEXT_LEN, prefix_len=spec.PREFIX_LEN,
            sep_token_id=0)
        self.cfg = cfg
        h = C.c_void_p()
        self._chk(self.lib.mellow_engine_create(C.byref(cfg), self.device, C.byref(h)))
        self.h = h
        self.finalized = False
        # "f32x3" (default, = the library's default and the mode bench.py reports): fp32-accurate GEMMs as exact 3-way bf16 splits
        # on the bf16 MFMA pipe.  "f32": exact fp32 MFMA GEMMs.  "fp8": BASELINE config 5 (e4m3 GEMMs, not bit-exact).
        # The parity suite runs "f32x3" and "f32" with the same tolerances and exact tokens.  MELLOW_PRECISION overrides the default.
        precision = precision or os.environ.get("MELLOW_PRECISION") or DEFAULT_PRECISION
        if precision not in ("f32", "fp8", "f32x3"):
            raise ValueError(f"unknown precision {precision!r}")
        self.precision = precision
        self._chk(self.lib.mellow_engine_set_precision(self.h, {"f32": 0, "fp8": 1, "f32x3": 2}[precision]))
        # explicit configuration (include/mellow_hip.h: mellow_engine_set_option): the library reads no environment variable; the
        # A/B forms the tests and tools compare are selected here, by name, and show up in describe()["non_default"]
        for k, v in (options or {}).items():
            self.set_option(k, v)

    def set_option(self, key: str, value) -> None:
        self._chk(self.lib.mellow_engine_set_option(self.h, key.encode(), str(int(value)).encode()))

    def describe(self) -> dict:
        """the engine's resolved configuration (mellow_engine_describe): precision, every option with value / default, ABI"""
        import json
        n = int(self.lib.mellow_engine_describe(self.h, None, 0))
        buf = C.create_string_buffer(n)
        self.lib.mellow_engine_describe(self.h, buf, n)
        return json.loads(buf.value.decode())

    # ---- errors ------------------------------------------------------------------------------------
    def _chk(self, rc: int):
        if rc != 0:
            msg = self.lib.mellow_last_error().decode("utf-8", "replace")
            if msg.startswith("index out of range in self"):      # the reference's embedding lookup raises IndexError with this text
                raise IndexError(msg)
            raise EngineError(msg)

    def close(self):
        if getattr(self, "h", None):
            for f in getattr(self, "_forks", []):
                f.close()
            self.lib.mellow_engine_destroy(self.h)
            self.h = None

    def fork(self) -> "Engine":
        """Another execution context on the same device sharing this engine's weights (mellow_engine_fork): own stream, KV pages,
        workspaces and graphs; calls on the two objects may overlap from different threads.  Closed with (or before) its parent."""
        if not self.finalized:
            raise EngineError("fork needs a loaded engine")
        c = object.__new__(Engine)
        c.lib, c.lm, c.device, c.tdev, c.cfg, c.precision, c.finalized = self.lib, self.lm, self.device, self.tdev, self.cfg, self.precision, True
        h = C.c_void_p()
        self._chk(self.lib.mellow_engine_fork(self.h, C.byref(h)))
        c.h = h
        c._parent = self              # keeps the weights alive
        self._forks = getattr(self, "_forks", []) + [c]
        return c

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights -----------------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        """Mirror of `model.load_state_dict` (reference wrapper.py:74-82): every tensor goes to the engine
        under its reference key; a leading 'module.' is stripped by the engine."""
        for k, v in sd.items():
            t = v.detach().cpu().contiguous()
            if t.dtype == torch.float32:
                dt = _F32
            elif t.dtype == torch.int64:
                dt = _I64
            elif t.dtype == torch.int32:
                dt = _I32
            else:
                t, dt = t.float(), _F32
            shape = (C.c_int64 * max(1, t.dim()))(*t.shape)
            rc = self.lib.mellow_engine_load_tensor(self.h, k.encode(), C.c_void_p(t.data_ptr()), shape, t.dim(), dt)
            if rc != 0 and strict:
                self._chk(rc)
        cos, sin = hf_rope_tables(self.cfg.max_positions, self.lm.head_dim, self.lm.rope_theta)
        for name, arr in (("mellow.rope_cos", cos), ("mellow.rope_sin", sin)):
            shape = (C.c_int64 * 2)(*arr.shape)
            self._chk(self.lib.mellow_engine_load_tensor(self.h, name.encode(), arr.ctypes.data_as(C.c_void_p), shape, 2, _F32))
        self._chk(self.lib.mellow_engine_finalize(self.h))
        self.finalized = True

    def required_keys(self):
        n = self.lib.mellow_engine_num_required()
        return [self.lib.mellow_engine_required_key(i).decode() for i in range(n)]

    # ---- helpers -----------------------------------------------------------------------------------
    def _f32(self, x) -> torch.Tensor:
        t = torch.as_tensor(x)
        return t.to(device=self.tdev, dtype=torch.float32).contiguous()

    def _i32(self, x) -> torch.Tensor:
        t = torch.as_tensor(x)
        return t.to(device=self.tdev, dtype=torch.int32).contiguous()

    def _ids(self, x) -> torch.Tensor:
        """token ids -> int32 on the device; ids outside the vocabulary raise like the reference's embedding lookup
        (`lm.model.embed_tokens`, decoder.py:47 / wrapper.py:237) instead of reaching a kernel."""
        t = torch.as_tensor(x)
        if t.numel() and (int(t.min()) < 0 or int(t.max()) >= self.lm.vocab_size):
            raise IndexError("index out of range in self")
        return t.to(device=self.tdev, dtype=torch.int32).contiguous()

    def _prompt_ids(self, x) -> torch.Tensor:
        """prompt ids -> int32 on the device with NO torch kernel and NO synchronisation when they already are device int32 (the
        timed path of bench.py): their range is checked on the device by prefix_assemble_kernel, which flags the call's error word;
        the C call then fails with the reference's IndexError text.  Host arrays (what a tokenizer returns) are checked right here
        in numpy; a device tensor of another dtype is clamped to [-1, vocab] before it is narrowed, so an out-of-range 64-bit id
        cannot alias a valid 32-bit one."""
        t = torch.as_tensor(x)
        if t.device.type == "cpu":
            a = t.numpy()
            if a.size and (int(a.min()) < 0 or int(a.max()) >= self.lm.vocab_size):
                raise IndexError("index out of range in self")
        elif t.dtype != torch.int32:
            t = t.clamp(-1, self.lm.vocab_size)
        return t.to(device=self.tdev, dtype=torch.int32).contiguous()

    def _sync_inputs(self):
        """The engine runs on its own non-blocking HIP stream: device tensors produced by still-running torch kernels
        (resample / tile / cat on torch's current stream) must be complete before their raw pointers cross the C ABI."""
        torch.cuda.current_stream(self.tdev).synchronize()

    def max_new_tokens_limit(self) -> int:
        """largest max_len the KV pages / RoPE tables of this engine can hold"""
        return int(self.cfg.max_positions) - spec.PREFIX_LEN

    # ---- hot path ----------------------------------------------------------------------------------
    def generate(self, audio1, audio2, input_ids, max_len: int, top_p: float = 0.8, temperature: float = 1.0,
                 stop_id: int = 0, ignore_stop: bool = False):
        """-> (tokens int32 [B, steps] on host, lengths [B], steps, first_token_ms)
        first_token_ms is measured from the C entry (inputs on the device); `last_first_token_host_ms` adds the time this
        call spent bringing host arrays to the device (latency from audio in HOST memory)."""
        import time
        t_in = time.perf_counter()
        a1, a2, ids = self._f32(audio1), self._f32(audio2), self._prompt_ids(input_ids)
        B, n = a1.shape
        assert a2.shape == a1.shape and ids.shape == (B, spec.TEXT_LEN), (a1.shape, a2.shape, ids.shape)
        out = torch.empty((B, max_len), dtype=torch.int32, device=self.tdev)
        self._sync_inputs()
        t_up = (time.perf_counter() - t_in) * 1e3
        lens = (C.c_int32 * B)()
        steps = C.c_int32(0)
        ftm = C.c_float(0.0)
        self._chk(self.lib.mellow_generate(self.h, _ptr(a1), _ptr(a2), n, _ptr(ids), B, int(max_len), float(top_p),
                                           float(temperature), int(stop_id), 1 if ignore_stop else 0, _ptr(out),
                                           lens, C.byref(steps), C.byref(ftm)))
        toks = out.cpu().numpy()[:, : steps.value]
        self.last_first_token_host_ms = t_up + float(ftm.value)
        return toks, np.asarray(list(lens), dtype=np.int32), int(steps.value), float(ftm.value)

    def stft_is_fft(self) -> bool:
        """the STFT runs as an FFT (f32x3 mode, windowed-DFT conv weights) instead of the DFT GEMM"""
        return bool(self.lib.mellow_stft_is_fft(self.h))

    def prefill_parts(self) -> int:
        """parts the f32x3 LM prefill runs as (2 = two half-batches on two streams MEASURED to overlap; 1 = one chain, also the
        fallback when this process's HIP runtime has no second hardware queue for the engine)"""
        return int(self.lib.mellow_prefill_parts(self.h))

    def last_row_repacks(self) -> int:
        """how often the last generate() call packed the still-running rows into fewer 32-row blocks"""
        return int(self.lib.mellow_last_row_repacks(self.h))

    def last_steps_enqueued(self) -> int:
        return int(self.lib.mellow_last_steps_enqueued(self.h))

    # ---- taps ----------------------------------------------------------------------------------------
    def logmel(self, wav, apply_bn: bool = False) -> torch.Tensor:
        w = self._f32(wav)
        n, ns = w.shape
        out = torch.empty((n, spec.frames_for(ns), spec.MEL_BINS), dtype=torch.float32, device=self.tdev)
        self._sync_inputs()
        self._chk(self.lib.mellow_logmel(self.h, _ptr(w), n, ns, 1 if apply_bn else 0, _ptr(out)))
        return out

    def encode(self, wav) -> torch.Tensor:
        w = self._f32(wav)
        n, ns = w.shape
        out = torch.empty((n, spec.AUDIO_ROWS, spec.D_PROJ), dtype=torch.float32, device=self.tdev)
        self._sync_inputs()
        self._chk(self.lib.mellow_encode(self.h, _ptr(w), n, ns, _ptr(out)))
        return out

    def prefix(self, audio1, audio2, input_ids) -> torch.Tensor:
        a1, a2, ids = self._f32(audio1), self._f32(audio2), self._prompt_ids(input_ids)
        B, n = a1.shape
        out = torch.empty((B, spec.PREFIX_LEN, spec.D_PROJ), dtype=torch.float32, device=self.tdev)
        self._sync_inputs()
        self._chk(self.lib.mellow_prefix(self.h, _ptr(a1), _ptr(a2), n, _ptr(ids), B, _ptr(out)))
        return out

    def lm_prefill(self, prefix, reserve: int = 64) -> torch.Tensor:
        p = self._f32(prefix)
        B, T, H = p.shape
        out = torch.empty((B, self.lm.vocab_size), dtype=torch.float32, device=self.tdev)
        self._sync_inputs()
        self._chk(self.lib.mellow_lm_prefill(self.h, _ptr(p), B, T, int(reserve), _ptr(out)))
        return out

    def lm_decode_step(self, token_ids) -> torch.Tensor:
        t = self._ids(token_ids).reshape(-1)
        out = torch.empty((t.shape[0], self.lm.vocab_size), dtype=torch.float32, device=self.tdev)
        self._sync_inputs()
        self._chk(self.lib.mellow_lm_decode_step(self.h, _ptr(t), _ptr(out)))
        return out

    def embed_tokens(self, token_ids) -> torch.Tensor:
        """`lm.model.embed_tokens(ids)` (reference decoder.py:47, wrapper.py:237): (...,) ids -> (..., hidden) on the device"""
        t = self._ids(token_ids)
        out = torch.empty(tuple(t.shape) + (self.lm.hidden_size,), dtype=torch.float32, device=self.tdev)
        self._sync_inputs()
        self._chk(self.lib.mellow_embed_tokens(self.h, _ptr(t), t.numel(), _ptr(out)))
        return out

    def lm_forward_logits(self, embeds, from_pos: int = 0) -> torch.Tensor:
        """`lm(inputs_embeds=embeds).logits[:, from_pos:]` (reference decoder.py:89): (B, T, hidden) -> (B, T - from_pos, vocab)"""
        p = self._f32(embeds)
        B, T, H = p.shape
        out = torch.empty((B, T - int(from_pos), self.lm.vocab_size), dtype=torch.float32, device=self.tdev)
        self._sync_inputs()
        self._chk(self.lib.mellow_lm_forward_logits(self.h, _ptr(p), B, T, int(from_pos), _ptr(out)))
        return out

    def forward(self, audio1, audio2, input_ids, answer_ids, from_pos: int = 0) -> torch.Tensor:
        """The training-time forward of the reference as inference arithmetic (`Mellow.forward`, mellow.py:89-98): logits of
        the sequence [audio1 | sep | audio2 | sep | prompt | answer] at every position >= from_pos.  The reference returns the
        HF output object; `.logits` of it is what this returns (no labels, no loss: decoder.py:84-89 passes labels=None)."""
        prefix = self.prefix(audio1, audio2, input_ids)
        ans = self.embed_tokens(answer_ids)
        return self.lm_forward_logits(torch.cat((prefix, ans), 1), from_pos)

    # -- the reference model object's call surface (wrapper.model is an nn.Module there: mellow.py:70-109) -------------------
    def generate_prefix_inference(self, input_dict):
        """`Mellow.generate_prefix_inference(input_dict)` (mellow.py:100-109): (prefix, None, None) -- the reference's second
        and third values are the encoder's output dicts, which the generation path never reads (wrapper.py:233)"""
        return self.prefix(input_dict["audio1"], input_dict["audio2"], input_dict["input"]["input_ids"]), None, None

    def __call__(self, input_dict):
        """`model(input_dict)` (mellow.py:89-98): an object whose `.logits` is (B, prefix + answer, vocab), like the HF
        CausalLMOutput the reference returns with labels=None (decoder.py:89: loss is None)"""
        from types import SimpleNamespace
        logits = self.forward(input_dict["audio1"], input_dict["audio2"], input_dict["input"]["input_ids"],
                              input_dict["answer"]["input_ids"])
        return SimpleNamespace(logits=logits, loss=None)

    def argmax(self, logits) -> torch.Tensor:
        l = self._f32(logits)
        out = torch.empty((l.shape[0],), dtype=torch.int32, device=self.tdev)
        self._sync_inputs()
        self._chk(self.lib.mellow_argmax(self.h, _ptr(l), l.shape[0], _ptr(out)))
        return out

    def resample(self, wav, orig_freq: int, new_freq: int) -> torch.Tensor:
        """(n, n_in) -> (n, ceil(new*n_in/orig)) on the device: the A0 resampler (twin of mellow_amd.audio.resample)."""
        w = self._f32(wav)
        if w.dim() == 1:
            w = w[None]
        n, n_in = w.shape
        n_out = C.c_int64(0)
        self._sync_inputs()
        self._chk(self.lib.mellow_resample(self.h, _ptr(w), n, n_in, int(orig_freq), int(new_freq), None, 0, C.byref(n_out)))
        out = torch.empty((n, n_out.value), dtype=torch.float32, device=self.tdev)
        self._chk(self.lib.mellow_resample(self.h, _ptr(w), n, n_in, int(orig_freq), int(new_freq), _ptr(out), n_out.value,
                                           C.byref(n_out)))
        return out

    def debug_gemm_f32(self, A: torch.Tensor, W: torch.Tensor, mode: int = 0, iters: int = 0):
        """C = A . W^T through the exact fp32 MFMA kernel (mode 0) or a bf16x3 split kernel (9 / 6: pre-split rows; 16: the fused
        kernel the f32x3 mode runs); host tensors."""
        A = A.detach().cpu().contiguous().float()
        W = W.detach().cpu().contiguous().float()
        (M, K), (N, K2) = A.shape, W.shape
        assert K == K2
        out = torch.empty((M, N), dtype=torch.float32)
        ms = (C.c_float * 2)()
        self._chk(self.lib.mellow_debug_gemm_f32(self.h, int(mode), C.c_void_p(A.data_ptr()), M, K, C.c_void_p(W.data_ptr()), N,
                                                 C.c_void_p(out.data_ptr()), int(iters), ms if iters > 0 else None))
        return out, ((ms[0], ms[1]) if iters > 0 else None)

    def debug_gemm_fp8(self, A: torch.Tensor, W: torch.Tensor, iters: int = 0):
        """fp8 mode quantisation tap: (C = A . W^T through the e4m3 GEMM, (quant_ms, gemm_ms) or None); host tensors."""
        A = A.detach().cpu().contiguous().float()
        W = W.detach().cpu().contiguous().float()
        (M, K), (N, K2) = A.shape, W.shape
        assert K == K2
        out = torch.empty((M, N), dtype=torch.float32)
        ms = (C.c_float * 2)()
        self._chk(self.lib.mellow_debug_gemm_fp8(self.h, C.c_void_p(A.data_ptr()), M, K, C.c_void_p(W.data_ptr()), N,
                                                 C.c_void_p(out.data_ptr()), int(iters), ms if iters > 0 else None))
        return out, ((ms[0], ms[1]) if iters > 0 else None)

    def debug_dec_head(self, x: torch.Tensor, act_fp8: bool = False) -> torch.Tensor:
        """The decode step's lm_head kernel on the rows x [B, hidden] (device tensor) -> logits [B, vocab]."""
        x = x.to(self.tdev).contiguous().float()
        out = torch.empty((x.shape[0], self.lm.vocab_size), dtype=torch.float32, device=self.tdev)
        self._chk(self.lib.mellow_debug_dec_head(self.h, _ptr(x), x.shape[0], 1 if act_fp8 else 0, _ptr(out)))
        return out

    def enable_taps(self, on: bool = True):
        self._chk(self.lib.mellow_debug_enable_taps(self.h, 1 if on else 0))

    def tap(self, name: str) -> torch.Tensor:
        n = C.c_int64(0)
        self._chk(self.lib.mellow_debug_tap(self.h, name.encode(), None, 0, C.byref(n)))
        out = torch.empty((n.value,), dtype=torch.float32, device=self.tdev)
        self._chk(self.lib.mellow_debug_tap(self.h, name.encode(), _ptr(out), n.value, C.byref(n)))
        return out

    # ---- measurement -----------------------------------------------------------------------------------
    def prof_enable(self, on: bool = True):
        self._chk(self.lib.mellow_prof_enable(self.h, 1 if on else 0))

    def prof_reset(self):
        self._chk(self.lib.mellow_prof_reset(self.h))

    def prof_report(self):
        out = {}
        for i in range(self.lib.mellow_prof_num_families()):
            n, ms, fl, by = C.c_int64(0), C.c_double(0), C.c_double(0), C.c_double(0)
            self._chk(self.lib.mellow_prof_get(self.h, i, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)))
            out[self.lib.mellow_prof_family_name(i).decode()] = {
                "launches": n.value, "ms": ms.value, "flops": fl.value, "bytes": by.value}
        return out

    def last_phase_ms(self):
        a, b, c = C.c_float(0), C.c_float(0), C.c_float(0)
        self._chk(self.lib.mellow_last_phase_ms(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"encode_ms": a.value, "prefill_ms": b.value, "decode_ms": c.value}

    def set_graph(self, on: bool):
        self._chk(self.lib.mellow_set_graph(self.h, 1 if on else 0))


# ---- host-only helpers (usable without a GPU) ------------------------------------------------------------
def host_window_map(R: int, shift: int) -> np.ndarray:
    lib = load_library()
    out = (C.c_int32 * (R * R))()
    if lib.mellow_host_window_map(R, shift, out) != 0:
        raise EngineError(lib.mellow_last_error().decode())
    return np.asarray(list(out), dtype=np.int32)


def host_pack_weight(w: np.ndarray, npad: int = 128) -> np.ndarray:
    lib = load_library()
    w = np.ascontiguousarray(w, dtype=np.float32)
    N, K = w.shape
    NP, KP = (N + npad - 1) // npad * npad, (K + 31) // 32 * 32
    out = np.empty((NP // 32, KP // 8, 64, 4), dtype=np.float32)
    rc = lib.mellow_host_pack_weight(w.ctypes.data_as(C.POINTER(C.c_float)), N, K, npad,
                                     out.ctypes.data_as(C.POINTER(C.c_float)), out.size)
    if rc != 0:
        raise EngineError(lib.mellow_last_error().decode())
    return out
