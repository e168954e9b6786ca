"""Static description of the Mellow v0 / v0_s model: encoder constants, LM hyper-parameters and
the checkpoint (`state_dict`) key layout the engine consumes.

Everything here is data about the reference's model, cited by file:line:
  * encoder constants            reference mellow/model/config.py:1-10
  * Swin geometry                reference mellow/model/htsat.py:599-606 (spec 256, patch 4, embed 96,
                                 depths [2,2,6,2], heads [4,8,16,32], window 8, mlp_ratio 4)
  * checkpoint key families      SURVEY.md §8b (479 entries, probe of the reference `state_dict`)
"""
from __future__ import annotations

import os
from collections import OrderedDict
from dataclasses import dataclass, field

import yaml

# ---- audio front-end (reference mellow/model/config.py:1-10) ---------------------------------
SAMPLE_RATE = 32000
WINDOW_SIZE = 1024
HOP_SIZE = 320
MEL_BINS = 64
FMIN = 50
FMAX = 14000
N_FREQ = WINDOW_SIZE // 2 + 1  # 513

# ---- HTSAT-Swin geometry (reference htsat.py:599-606, 680-696) ----------------------------------
SPEC_SIZE = 256
FREQ_RATIO = SPEC_SIZE // MEL_BINS  # 4
PATCH = 4
EMBED_DIM = 96
DEPTHS = (2, 2, 6, 2)
NUM_HEADS = (4, 8, 16, 32)
WINDOW = 8
WIN_TOKENS = WINDOW * WINDOW  # 64
HEAD_DIM = 24
NUM_CLASSES = 527
ENC_OUT = 768
D_PROJ = 576
# long-audio crop constants hard-coded in the reference (htsat.py:912-915)
LONG_CROP = 689
LONG_HOP = 344

# tokens / channels per stage
STAGE_RES = tuple(SPEC_SIZE // PATCH // (2 ** i) for i in range(4))  # 64, 32, 16, 8
STAGE_DIM = tuple(EMBED_DIM * (2 ** i) for i in range(4))            # 96, 192, 384, 768

# ---- prefix layout (reference decoder.py:36-55, v0.yaml:9,16) -------------------------------------
AUDIO_ROWS = 129            # 1 latent + 128 pooled frame rows (decoder.py:14-18)
TEXT_LEN = 129
PREFIX_LEN = 2 * AUDIO_ROWS + 2 + TEXT_LEN  # 389
DISTINCT_ROWS = 33          # 1 latent + 32 distinct framewise rows

_PKG = os.path.dirname(os.path.abspath(__file__))


@dataclass
class LMConfig:
    vocab_size: int = 49152
    hidden_size: int = 576
    intermediate_size: int = 1536
    num_hidden_layers: int = 30
    num_attention_heads: int = 9
    num_key_value_heads: int = 3
    head_dim: int = 64
    rms_norm_eps: float = 1e-5
    rope_theta: float = 100000.0
    max_position_embeddings: int = 8192
    tie_word_embeddings: bool = True
    bos_token_id: int = 0
    eos_token_id: int = 0

    @staticmethod
    def load(path: str | None = None) -> "LMConfig":
        path = path or os.path.join(_PKG, "config", "lm_smollm2_135m.yaml")
        with open(path, "r") as f:
            d = yaml.safe_load(f)
        return LMConfig(**d)


def frames_for(n_samples: int) -> int:
    """STFT frame count with center=True (Appendix B of SURVEY.md): n/hop + 1."""
    return n_samples // HOP_SIZE + 1


def long_crop_positions(n_frames: int):
    """Crop starts of the eval long-audio branch (reference htsat.py:916)."""
    return list(range(0, n_frames - LONG_CROP - 1, LONG_HOP))


# ---- checkpoint key layout -------------------------------------------------------------------------
ENC = "audio_encoder.base.htsat."
C2L = "audio_encoder.base.c2l."
PROJ = "audio_encoder.projection."
LM = "caption_decoder.lm."


def state_dict_layout(lm: LMConfig | None = None) -> "OrderedDict[str, tuple[tuple[int, ...], str]]":
    """name -> (shape, dtype) for every entry of the reference `state_dict`.

    dtype is 'f32' or 'i64' (the reference stores index buffers as int64)."""
    lm = lm or LMConfig()
    L: "OrderedDict[str, tuple[tuple[int, ...], str]]" = OrderedDict()

    def f(name, *shape):
        L[name] = (tuple(shape), "f32")

    def i(name, *shape):
        L[name] = (tuple(shape), "i64")

    f(ENC + "spectrogram_extractor.stft.conv_real.weight", N_FREQ, 1, WINDOW_SIZE)
    f(ENC + "spectrogram_extractor.stft.conv_imag.weight", N_FREQ, 1, WINDOW_SIZE)
    f(ENC + "logmel_extractor.melW", N_FREQ, MEL_BINS)
    f(ENC + "bn0.weight", MEL_BINS)
    f(ENC + "bn0.bias", MEL_BINS)
    f(ENC + "bn0.running_mean", MEL_BINS)
    f(ENC + "bn0.running_var", MEL_BINS)
    i(ENC + "bn0.num_batches_tracked")
    f(ENC + "patch_embed.proj.weight", EMBED_DIM, 1, PATCH, PATCH)
    f(ENC + "patch_embed.proj.bias", EMBED_DIM)
    f(ENC + "patch_embed.norm.weight", EMBED_DIM)
    f(ENC + "patch_embed.norm.bias", EMBED_DIM)
    for s in range(4):
        C, nH, R = STAGE_DIM[s], NUM_HEADS[s], STAGE_RES[s]
        nW = (R // WINDOW) ** 2
        for b in range(DEPTHS[s]):
            p = f"{ENC}layers.{s}.blocks.{b}."
            shifted = (b % 2 == 1) and R > WINDOW
            if shifted:
                f(p + "attn_mask", nW, WIN_TOKENS, WIN_TOKENS)
            f(p + "norm1.weight", C)
            f(p + "norm1.bias", C)
            f(p + "attn.relative_position_bias_table", (2 * WINDOW - 1) ** 2, nH)
            i(p + "attn.relative_position_index", WIN_TOKENS, WIN_TOKENS)
            f(p + "attn.qkv.weight", 3 * C, C)
            f(p + "attn.qkv.bias", 3 * C)
            f(p + "attn.proj.weight", C, C)
            f(p + "attn.proj.bias", C)
            f(p + "norm2.weight", C)
            f(p + "norm2.bias", C)
            f(p + "mlp.fc1.weight", 4 * C, C)
            f(p + "mlp.fc1.bias", 4 * C)
            f(p + "mlp.fc2.weight", C, 4 * C)
            f(p + "mlp.fc2.bias", C)
        if s < 3:
            p = f"{ENC}layers.{s}.downsample."
            f(p + "reduction.weight", 2 * C, 4 * C)
            f(p + "norm.weight", 4 * C)
            f(p + "norm.bias", 4 * C)
    f(ENC + "norm.weight", ENC_OUT)
    f(ENC + "norm.bias", ENC_OUT)
    f(ENC + "tscam_conv.weight", NUM_CLASSES, ENC_OUT, 2, 3)
    f(ENC + "tscam_conv.bias", NUM_CLASSES)
    f(ENC + "head.weight", NUM_CLASSES, NUM_CLASSES)   # present in the ckpt, never used (htsat.py:710)
    f(ENC + "head.bias", NUM_CLASSES)
    f(C2L + "weight", ENC_OUT, NUM_CLASSES)
    f(C2L + "bias", ENC_OUT)
    f(PROJ + "linear1.weight", D_PROJ, ENC_OUT)
    f(PROJ + "linear2.weight", D_PROJ, D_PROJ)
    f(PROJ + "layer_norm.weight", D_PROJ)
    f(PROJ + "layer_norm.bias", D_PROJ)
    H, I = lm.hidden_size, lm.intermediate_size
    KV = lm.num_key_value_heads * lm.head_dim
    Q = lm.num_attention_heads * lm.head_dim
    f(LM + "model.embed_tokens.weight", lm.vocab_size, H)
    for l in range(lm.num_hidden_layers):
        p = f"{LM}model.layers.{l}."
        f(p + "self_attn.q_proj.weight", Q, H)
        f(p + "self_attn.k_proj.weight", KV, H)
        f(p + "self_attn.v_proj.weight", KV, H)
        f(p + "self_attn.o_proj.weight", H, Q)
        f(p + "mlp.gate_proj.weight", I, H)
        f(p + "mlp.up_proj.weight", I, H)
        f(p + "mlp.down_proj.weight", H, I)
        f(p + "input_layernorm.weight", H)
        f(p + "post_attention_layernorm.weight", H)
    f(LM + "model.norm.weight", H)
    f(LM + "lm_head.weight", lm.vocab_size, H)  # tied to embed_tokens (same storage in the reference)
    return L


# keys that exist in the checkpoint but the inference path never reads
UNUSED_KEYS = (
    ENC + "bn0.num_batches_tracked",
    ENC + "head.weight",
    ENC + "head.bias",
)
