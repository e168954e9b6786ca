"""CPU ORACLE for the Mellow inference hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a plain fp32 PyTorch-CPU restatement of the reference algorithm (soham97/mellow), written
against a raw `state_dict` (no nn.Module), one function per reference function, each citing the
reference file:line it follows.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import it, and only as the checker / the timed CPU baseline.  The product path
(`mellow_amd`) never imports it and fails loudly when the HIP library is missing.

It mirrors the reference op for op on purpose, including what the reference does wastefully:
DFT as two conv1d, two separate encoder passes, no KV cache (full re-forward of the 389+i token
sequence per generated token), logits for every position, sort-based top-p before arg-max.

PINNING:
  * pinned HERE against the imported reference (`/root/reference`, via tests/golden/make_golden.py,
    which loads the same synthetic state_dict strict into the reference modules) on every tap; the
    resulting vectors are committed under tests/golden/ and re-checked by tests/test_oracle_golden.py.
  * PARITY UNPINNED at the third-party boundaries the reference does not vendor and this container
    does not have: torchlibrosa 0.1.0 (STFT/log-mel op sequence restated from its published
    algorithm; its constants are checkpoint tensors so only the op order is restated),
    transformers 4.46.3 `LlamaForCausalLM` (checked against the installed transformers 5.15 Llama
    instead, same math), torchaudio 2.0.1's binary (its published resampling algorithm has its own
    independent fp64 oracle: oracle/resample_oracle.py), and the SmolLM2 tokenizer files.  The reference itself ships no tests or golden vectors.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

# Geometry constants of the reference encoder (reference mellow/model/config.py:1-10,
# mellow/model/htsat.py:599-606).  Restated here so the oracle has no dependency on the product package.
WINDOW_SIZE, HOP_SIZE, MEL_BINS = 1024, 320, 64
SPEC_SIZE, FREQ_RATIO, PATCH = 256, 4, 4
DEPTHS, NUM_HEADS, WINDOW = (2, 2, 6, 2), (4, 8, 16, 32), 8
LONG_CROP, LONG_HOP = 689, 344
ENC = "audio_encoder.base.htsat."
C2L = "audio_encoder.base.c2l."
PROJ = "audio_encoder.projection."
LM = "caption_decoder.lm."

SD = Dict[str, torch.Tensor]


# =====================================================================================================
# A1-A4: front-end
# =====================================================================================================
def stft_power(sd: SD, wav: torch.Tensor) -> torch.Tensor:
    """torchlibrosa Spectrogram(power=2) as used at reference htsat.py:647-649, called :864.
    (B,n) -> (B,1,frames,513): reflect-pad n_fft/2, two conv1d with the checkpoint's windowed-DFT
    weights, re^2+im^2."""
    x = wav[:, None, :]
    x = F.pad(x, (WINDOW_SIZE // 2, WINDOW_SIZE // 2), mode="reflect")
    real = F.conv1d(x, sd[ENC + "spectrogram_extractor.stft.conv_real.weight"], stride=HOP_SIZE)
    imag = F.conv1d(x, sd[ENC + "spectrogram_extractor.stft.conv_imag.weight"], stride=HOP_SIZE)
    real = real[:, None, :, :].transpose(2, 3)
    imag = imag[:, None, :, :].transpose(2, 3)
    return real ** 2 + imag ** 2


def logmel(sd: SD, power: torch.Tensor) -> torch.Tensor:
    """torchlibrosa LogmelFilterBank(ref=1, amin=1e-10, top_db=None), reference htsat.py:651-653, :865."""
    mel = torch.matmul(power, sd[ENC + "logmel_extractor.melW"])
    log_spec = 10.0 * torch.log10(torch.clamp(mel, min=1e-10, max=float("inf")))
    log_spec = log_spec - 10.0 * math.log10(max(1e-10, 1.0))
    return log_spec


def bn0(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """BatchNorm2d(64) in eval mode over the mel axis, reference htsat.py:657, :868-870."""
    x = x.transpose(1, 3)
    x = F.batch_norm(x, sd[ENC + "bn0.running_mean"], sd[ENC + "bn0.running_var"],
                     sd[ENC + "bn0.weight"], sd[ENC + "bn0.bias"], training=False, eps=1e-5)
    return x.transpose(1, 3)


def reshape_wav2img(x: torch.Tensor) -> torch.Tensor:
    """reference htsat.py:830-845: bicubic(align_corners) to 1024 frames, fold 4x256 time chunks on freq."""
    B, C, T, Fq = x.shape
    target_T = SPEC_SIZE * FREQ_RATIO
    target_F = SPEC_SIZE // FREQ_RATIO
    assert T <= target_T and Fq <= target_F, "the wav size should less than or equal to the swin input size"
    if T < target_T:
        x = F.interpolate(x, (target_T, x.shape[3]), mode="bicubic", align_corners=True)
    if Fq < target_F:
        x = F.interpolate(x, (x.shape[2], target_F), mode="bicubic", align_corners=True)
    x = x.permute(0, 1, 3, 2).contiguous()
    x = x.reshape(x.shape[0], x.shape[1], x.shape[2], FREQ_RATIO, x.shape[3] // FREQ_RATIO)
    x = x.permute(0, 1, 3, 2, 4).contiguous()
    x = x.reshape(x.shape[0], x.shape[1], x.shape[2] * x.shape[3], x.shape[4])
    return x


# =====================================================================================================
# A5-A9: Swin encoder body
# =====================================================================================================
def patch_embed(sd: SD, img: torch.Tensor) -> torch.Tensor:
    """reference htsat.py:86-116: Conv2d(1->96,k4,s4) -> flatten -> LayerNorm(96)."""
    assert img.shape[2] == SPEC_SIZE and img.shape[3] == SPEC_SIZE
    x = F.conv2d(img, sd[ENC + "patch_embed.proj.weight"], sd[ENC + "patch_embed.proj.bias"], stride=PATCH)
    x = x.flatten(2).transpose(1, 2)
    return F.layer_norm(x, (x.shape[-1],), sd[ENC + "patch_embed.norm.weight"], sd[ENC + "patch_embed.norm.bias"])


def window_partition(x: torch.Tensor, ws: int) -> torch.Tensor:
    """reference htsat.py:224-235."""
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)


def window_reverse(windows: torch.Tensor, ws: int, H: int, W: int) -> torch.Tensor:
    """reference htsat.py:238-251."""
    B = int(windows.shape[0] / (H * W / ws / ws))
    x = windows.view(B, H // ws, W // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


def window_attention(sd: SD, p: str, x: torch.Tensor, nH: int, mask: Optional[torch.Tensor]) -> torch.Tensor:
    """reference htsat.py:301-332 (the returned attention map is dropped: nothing consumes it when
    htsat_attn_heatmap is False, config.py:10)."""
    B_, N, C = x.shape
    qkv = F.linear(x, sd[p + "qkv.weight"], sd[p + "qkv.bias"])
    qkv = qkv.reshape(B_, N, 3, nH, C // nH).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = q * ((C // nH) ** -0.5)
    attn = q @ k.transpose(-2, -1)
    table = sd[p + "relative_position_bias_table"]
    index = sd[p + "relative_position_index"]
    bias = table[index.view(-1)].view(N, N, -1).permute(2, 0, 1).contiguous()
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = attn.view(B_ // nW, nW, nH, N, N) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, nH, N, N)
    attn = torch.softmax(attn, dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B_, N, C)
    return F.linear(x, sd[p + "proj.weight"], sd[p + "proj.bias"])


def swin_block(sd: SD, s: int, b: int, x: torch.Tensor, res: int) -> torch.Tensor:
    """reference htsat.py:414-455 (eval: DropPath is identity)."""
    p = f"{ENC}layers.{s}.blocks.{b}."
    B, L, C = x.shape
    ws = WINDOW
    shift = 0 if (b % 2 == 0) else WINDOW // 2
    if res <= ws:                       # htsat.py:368-371
        shift, ws = 0, res
    shortcut = x
    x = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"]).view(B, res, res, C)
    if shift > 0:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = window_partition(x, ws).view(-1, ws * ws, C)
    mask = sd[p + "attn_mask"] if shift > 0 else None
    aw = window_attention(sd, p + "attn.", xw, NUM_HEADS[s], mask).view(-1, ws, ws, C)
    x = window_reverse(aw, ws, res, res)
    if shift > 0:
        x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
    x = shortcut + x.view(B, res * res, C)
    h = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    h = F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
    h = F.gelu(h)                       # nn.GELU(): exact erf form (htsat.py:121,126)
    h = F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + h


def patch_merging(sd: SD, s: int, x: torch.Tensor, res: int) -> torch.Tensor:
    """reference htsat.py:478-499."""
    p = f"{ENC}layers.{s}.downsample."
    B, L, C = x.shape
    x = x.view(B, res, res, C)
    x = torch.cat([x[:, 0::2, 0::2, :], x[:, 1::2, 0::2, :], x[:, 0::2, 1::2, :], x[:, 1::2, 1::2, :]], -1)
    x = x.view(B, -1, 4 * C)
    x = F.layer_norm(x, (4 * C,), sd[p + "norm.weight"], sd[p + "norm.bias"])
    return F.linear(x, sd[p + "reduction.weight"])


def forward_features(sd: SD, img: torch.Tensor, taps: Optional[dict] = None) -> Dict[str, torch.Tensor]:
    """reference htsat.py:733-796 with enable_tscam=True (config.py:3)."""
    frames_num = img.shape[2]
    x = patch_embed(sd, img)
    if taps is not None:
        taps["patch"] = x
    res = SPEC_SIZE // PATCH
    for s in range(4):
        for b in range(DEPTHS[s]):
            x = swin_block(sd, s, b, x, res)
        if s < 3:
            x = patch_merging(sd, s, x, res)
            res //= 2
        if taps is not None:
            taps[f"stage{s}"] = x
    x = F.layer_norm(x, (x.shape[-1],), sd[ENC + "norm.weight"], sd[ENC + "norm.bias"])
    B, N, C = x.shape
    SF = frames_num // (2 ** (len(DEPTHS) - 1)) // PATCH
    ST = frames_num // (2 ** (len(DEPTHS) - 1)) // PATCH
    x = x.permute(0, 2, 1).contiguous().reshape(B, C, SF, ST)
    B, C, Fq, T = x.shape
    c_freq_bin = Fq // FREQ_RATIO
    x = x.reshape(B, C, Fq // c_freq_bin, c_freq_bin, T)
    x = x.permute(0, 1, 3, 2, 4).contiguous().reshape(B, C, c_freq_bin, -1)
    latent = torch.flatten(F.adaptive_avg_pool1d(torch.flatten(x, 2), 1), 1)
    x = F.conv2d(x, sd[ENC + "tscam_conv.weight"], sd[ENC + "tscam_conv.bias"], padding=(0, 1))
    x = torch.flatten(x, 2)                                            # B, 527, 32
    fpx_small = torch.sigmoid(x).permute(0, 2, 1).contiguous()          # B, 32, 527
    ratio = 8 * PATCH                                                   # htsat.py:780 + interpolate() :43-56
    fpx = fpx_small[:, :, None, :].repeat(1, 1, ratio, 1).reshape(B, fpx_small.shape[1] * ratio, -1)
    clip = torch.sigmoid(torch.flatten(F.adaptive_avg_pool1d(x, 1), 1))
    return {"framewise_output": fpx, "clipwise_output": clip, "latent_output": latent}


def htsat_forward(sd: SD, wav: torch.Tensor, taps: Optional[dict] = None) -> Dict[str, torch.Tensor]:
    """reference htsat.py:863-941, eval mode, infer_mode=False, enable_repeat_mode=False."""
    x = stft_power(sd, wav)
    if taps is not None:
        taps["power"] = x
    x = logmel(sd, x)
    if taps is not None:
        taps["logmel"] = x
    x = bn0(sd, x)
    if taps is not None:
        taps["logmel_bn"] = x
    if x.shape[2] > FREQ_RATIO * SPEC_SIZE:
        # long-audio branch (htsat.py:908-936): fixed 689-frame crops every 344 frames, outputs averaged
        outs = []
        for cur_pos in range(0, x.shape[2] - LONG_CROP - 1, LONG_HOP):
            tx = x[:, :, cur_pos:cur_pos + LONG_CROP, :].clone()        # crop_wav with spe_pos (htsat.py:818-827)
            outs.append(forward_features(sd, reshape_wav2img(tx)))
        out = {k: torch.zeros_like(outs[0][k]) for k in ("clipwise_output", "framewise_output", "latent_output")}
        for d in outs:
            for k in out:
                out[k] = out[k] + d[k]
        for k in out:
            out[k] = out[k] / len(outs)
        if taps is not None:
            taps["n_crops"] = len(outs)
        return out
    img = reshape_wav2img(x)
    if taps is not None:
        taps["img"] = img
    return forward_features(sd, img, taps)


# =====================================================================================================
# A11-A14: embedding, projection, downsample, prefix
# =====================================================================================================
def htsat_wrapper_forward(sd: SD, wav: torch.Tensor, taps: Optional[dict] = None) -> Dict[str, torch.Tensor]:
    """reference htsat.py:950-955: c2l(framewise) and cat(latent)."""
    out = htsat_forward(sd, wav, taps)
    oframe = F.linear(out["framewise_output"], sd[C2L + "weight"], sd[C2L + "bias"])
    out["embedding"] = torch.cat((out["latent_output"].unsqueeze(1), oframe), dim=1)
    return out


def projection(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """reference mellow.py:48-52 (dropout off in eval)."""
    e1 = F.linear(x, sd[PROJ + "linear1.weight"])
    e2 = F.linear(F.gelu(e1), sd[PROJ + "linear2.weight"])
    return F.layer_norm(e1 + e2, (e1.shape[-1],), sd[PROJ + "layer_norm.weight"], sd[PROJ + "layer_norm.bias"])


def audio_encoder(sd: SD, wav: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
    """reference mellow.py:64-68 -> projected (B,1025,576)."""
    out = htsat_wrapper_forward(sd, wav, taps)
    if taps is not None:
        taps["latent"] = out["latent_output"]
        taps["framewise"] = out["framewise_output"]
        taps["embedding"] = out["embedding"]
    pv = projection(sd, out["embedding"])
    if taps is not None:
        taps["projected"] = pv
    return pv


def downsample(x: torch.Tensor) -> torch.Tensor:
    """reference decoder.py:14-18."""
    clip_latent = x[:, 0, :].unsqueeze(1)
    pooled = F.avg_pool2d(x[:, 1:, :], kernel_size=(8, 1))
    return torch.cat((clip_latent, pooled), dim=1)


def embed_tokens(sd: SD, ids: torch.Tensor) -> torch.Tensor:
    return F.embedding(ids, sd[LM + "model.embed_tokens.weight"])


def generate_prefix_inference(sd: SD, audio1: torch.Tensor, audio2: torch.Tensor, input_ids: torch.Tensor,
                              taps: Optional[dict] = None) -> torch.Tensor:
    """reference mellow.py:100-108 + decoder.py:36-55 (SmolLM2 branch: sep = embedding of id 0)."""
    t1 = {} if taps is not None else None
    e1 = audio_encoder(sd, audio1, t1)
    e2 = audio_encoder(sd, audio2, None)
    a1 = downsample(e1).contiguous()
    a2 = downsample(e2).contiguous()
    dtext = embed_tokens(sd, input_ids).contiguous()
    sep = embed_tokens(sd, torch.tensor([0])).unsqueeze(0).repeat(dtext.shape[0], 1, 1)
    if taps is not None:
        taps.update(t1)
        taps["audio1_ds"] = a1
        taps["audio2_ds"] = a2
    return torch.cat((a1, sep, a2, sep, dtext), dim=1)


# =====================================================================================================
# A15: Llama forward (transformers LlamaForCausalLM math; see module docstring for the pin)
# =====================================================================================================
class LMParams:
    def __init__(self, n_layers=30, n_heads=9, n_kv=3, head_dim=64, eps=1e-5, theta=100000.0):
        self.n_layers, self.n_heads, self.n_kv, self.head_dim, self.eps, self.theta = \
            n_layers, n_heads, n_kv, head_dim, eps, theta


def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """LlamaRMSNorm: fp32, x * rsqrt(mean(x^2)+eps), then weight * x."""
    var = x.pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps))


def rope_tables(T: int, head_dim: int, theta: float):
    """LlamaRotaryEmbedding (default rope): inv_freq = theta^(-2i/d); emb = cat(freqs, freqs)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    pos = torch.arange(T, dtype=torch.float32)
    freqs = (inv_freq[None, :, None].float() @ pos[None, None, :].float()).transpose(1, 2)[0]  # T, d/2
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def _rotate_half(x):
    x1 = x[..., : x.shape[-1] // 2]
    x2 = x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def llama_forward(sd: SD, lm: LMParams, inputs_embeds: torch.Tensor, last_only: bool = False) -> torch.Tensor:
    """LlamaForCausalLM.forward(inputs_embeds=...) without cache, as the reference calls it at
    wrapper.py:217: causal mask, position_ids = arange(T), eager softmax(QK^T*scale + mask) in fp32,
    GQA by repeat_kv, logits for all T positions (or only the last when last_only, a test shortcut)."""
    B, T, H = inputs_embeds.shape
    hd, nh, nkv = lm.head_dim, lm.n_heads, lm.n_kv
    cos, sin = rope_tables(T, hd, lm.theta)
    cos, sin = cos[None, None], sin[None, None]
    causal = torch.full((T, T), float("-inf")).triu(1)
    h = inputs_embeds
    for l in range(lm.n_layers):
        p = f"{LM}model.layers.{l}."
        x = rms_norm(h, sd[p + "input_layernorm.weight"], lm.eps)
        q = F.linear(x, sd[p + "self_attn.q_proj.weight"]).view(B, T, nh, hd).transpose(1, 2)
        k = F.linear(x, sd[p + "self_attn.k_proj.weight"]).view(B, T, nkv, hd).transpose(1, 2)
        v = F.linear(x, sd[p + "self_attn.v_proj.weight"]).view(B, T, nkv, hd).transpose(1, 2)
        q = q * cos + _rotate_half(q) * sin
        k = k * cos + _rotate_half(k) * sin
        rep = nh // nkv
        k = k[:, :, None].expand(B, nkv, rep, T, hd).reshape(B, nh, T, hd)
        v = v[:, :, None].expand(B, nkv, rep, T, hd).reshape(B, nh, T, hd)
        att = torch.matmul(q, k.transpose(2, 3)) * (hd ** -0.5) + causal
        att = torch.softmax(att, dim=-1, dtype=torch.float32)
        o = torch.matmul(att, v).transpose(1, 2).contiguous().reshape(B, T, nh * hd)
        h = h + F.linear(o, sd[p + "self_attn.o_proj.weight"])
        x = rms_norm(h, sd[p + "post_attention_layernorm.weight"], lm.eps)
        m = F.silu(F.linear(x, sd[p + "mlp.gate_proj.weight"])) * F.linear(x, sd[p + "mlp.up_proj.weight"])
        h = h + F.linear(m, sd[p + "mlp.down_proj.weight"])
    h = rms_norm(h, sd[LM + "model.norm.weight"], lm.eps)
    if last_only:
        h = h[:, -1:, :]
    return F.linear(h, sd[LM + "lm_head.weight"])


# =====================================================================================================
# A16: the generation loop
# =====================================================================================================
def generate_batch(sd: SD, lm: LMParams, prefix: torch.Tensor, entry_length: int, top_p: float,
                   temperature: float, stop_token_index: int, last_only: bool = True,
                   record: Optional[dict] = None) -> torch.Tensor:
    """reference wrapper.py:197-249 (SmolLM2 branch).  Returns int64 tokens (B, n_steps).

    `last_only=True` computes only the last position's logits (identical values; the reference
    computes all positions and slices [:, -1, :] at wrapper.py:218).  `record`, if given, receives the
    per-step last-position logits BEFORE the top-p filter (list under 'logits') and the top-2 gaps."""
    generated = prefix
    tokens = None
    filter_value = -float("inf")
    for _ in range(entry_length):
        logits = llama_forward(sd, lm, generated, last_only=last_only)
        logits = logits[:, -1, :] / (temperature if temperature > 0 else 1.0)
        if record is not None:
            record.setdefault("logits", []).append(logits.clone())
        sorted_logits, sorted_indices = torch.sort(logits, descending=True)
        cumulative_probs = torch.cumsum(F.softmax(sorted_logits, dim=-1), dim=-1)
        remove = cumulative_probs > top_p
        remove[..., 1:] = remove[..., :-1].clone()
        remove[..., 0] = 0
        for k in range(len(remove)):
            logits[k, sorted_indices[k][remove[k]]] = filter_value
        next_token = torch.argmax(logits, -1).unsqueeze(1)
        next_embed = embed_tokens(sd, next_token)
        tokens = next_token if tokens is None else torch.cat((tokens, next_token), dim=1)
        generated = torch.cat((generated, next_embed), dim=1)
        if ((tokens == stop_token_index).sum(dim=-1) > 0).all():
            break
    return tokens


def generate_tokens(sd: SD, lm: LMParams, audio1, audio2, input_ids, max_len: int, top_p: float = 0.8,
                    temperature: float = 1.0, stop_id: int = 0, record: Optional[dict] = None) -> torch.Tensor:
    """generate() minus host I/O and detokenisation (reference wrapper.py:277-286)."""
    with torch.no_grad():
        prefix = generate_prefix_inference(sd, torch.as_tensor(audio1), torch.as_tensor(audio2),
                                           torch.as_tensor(input_ids, dtype=torch.int64))
        return generate_batch(sd, lm, prefix, max_len, top_p, temperature, stop_id, record=record)


def cut_at_stop(tokens: torch.Tensor, stop_id: int) -> List[List[int]]:
    """Token-level equivalent of `.decode(x).split('<|endoftext|>')[0]` (wrapper.py:254)."""
    out = []
    for row in tokens.tolist():
        out.append(row[: row.index(stop_id)] if stop_id in row else row)
    return out
