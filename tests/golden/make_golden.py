#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference (`/root/reference`,
imported in this container) on the build's seeded synthetic checkpoint + synthetic inputs.

Run (build container only; /root/reference does not exist on the GPU box):
    python tests/golden/make_golden.py            # writes tests/golden/*.npz, prints oracle-vs-reference diffs

Recipe:
  1. put tests/golden/ref_shims on sys.path (torchlibrosa / torchaudio / importlib_resources are not
     installed here; see ref_shims/README.md — parity unpinned at those boundaries);
  2. monkeypatch AutoModelForCausalLM.from_pretrained (reference decoder.py:25, a network call) to build
     a random-init LlamaForCausalLM with the SmolLM2-135M hyper-parameters of
     mellow_amd/config/lm_smollm2_135m.yaml;
  3. construct `Mellow(...)` exactly as reference wrapper.py:66-73, `load_state_dict(strict=True)` the
     synthetic checkpoint (mellow_amd.synth.make_state_dict), `.eval()`;
  4. drive `Mellow.generate_prefix_inference` (mellow.py:100-108) and the unmodified
     `MellowWrapper._generate_batch` (wrapper.py:197-256) on an instance made with `__new__`
     (skipping the hub downloads of wrapper.py:41-42,84) with a stub tokenizer.

The vectors are DATA (inputs are regenerated from seeds; outputs are stored, sub-sampled where large).
No reference source text is stored.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, "ref_shims"))
sys.path.insert(0, REF)

from mellow_amd import spec, synth  # noqa: E402
from oracle import mellow_oracle as O  # noqa: E402

SEED = 0
N_STEPS = 12
SUB_VOCAB = np.arange(0, 49152, 96)  # 512 logits kept per step


def build_reference(sd):
    import transformers
    from transformers import LlamaConfig, LlamaForCausalLM

    lmc = spec.LMConfig.load()

    def fake_from_pretrained(name, *a, **k):
        cfg = LlamaConfig(vocab_size=lmc.vocab_size, hidden_size=lmc.hidden_size,
                          intermediate_size=lmc.intermediate_size, num_hidden_layers=lmc.num_hidden_layers,
                          num_attention_heads=lmc.num_attention_heads, num_key_value_heads=lmc.num_key_value_heads,
                          max_position_embeddings=lmc.max_position_embeddings, rms_norm_eps=lmc.rms_norm_eps,
                          rope_theta=lmc.rope_theta, tie_word_embeddings=lmc.tie_word_embeddings,
                          hidden_act="silu", attention_bias=False, mlp_bias=False,
                          bos_token_id=lmc.bos_token_id, eos_token_id=lmc.eos_token_id)
        return LlamaForCausalLM(cfg)

    transformers.AutoModelForCausalLM.from_pretrained = staticmethod(fake_from_pretrained)
    import mellow.model.decoder as ref_decoder
    ref_decoder.AutoModelForCausalLM.from_pretrained = staticmethod(fake_from_pretrained)
    from mellow.model.model import get_model_class
    import yaml
    with open(os.path.join(REF, "mellow", "config", "v0.yaml")) as f:
        cfg = yaml.safe_load(f)
    Model = get_model_class(cfg["model"]["model_type"])
    model = Model(audioenc_name=cfg["model"]["encoder"]["audioenc_name"], d_in=cfg["model"]["encoder"]["out_emb"],
                  text_decoder=cfg["model"]["decoder"]["text_decoder"],
                  prefix_length=cfg["model"]["decoder"]["prefix_length"], d_out=cfg["model"]["encoder"]["d_proj"])
    missing = model.load_state_dict(sd, strict=True)
    model.eval()
    n_params = sum(p.numel() for p in model.parameters())
    print(f"reference model built: {n_params} parameters, {len(model.state_dict())} state_dict entries; {missing}")
    assert n_params == 167020951, n_params
    return model, cfg


class StubTokenizer:
    def __init__(self, stop_id):
        self.stop_id = stop_id

    def encode(self, s):
        return [self.stop_id]

    def decode(self, ids):
        # token-level stand-in: one "word" per id, stop id rendered as the stop string
        return " ".join("<|endoftext|>" if int(i) == self.stop_id else f"t{int(i)}" for i in np.atleast_1d(ids))


def ref_generate_tokens(model, prefix, steps, stop_id, top_p=0.8, temperature=1.0):
    """Run the reference's own loop and capture tokens + per-step last-position logits via a hook."""
    from mellow.wrapper import MellowWrapper
    w = MellowWrapper.__new__(MellowWrapper)
    w.model = model
    w.tokenizer = StubTokenizer(stop_id)
    logits_log = []
    h = model.caption_decoder.lm.register_forward_hook(
        lambda m, i, o: logits_log.append(o.logits[:, -1, :].detach().clone()))
    try:
        strings = w._generate_batch(embed=prefix, entry_length=steps, top_p=top_p, temperature=temperature)
    finally:
        h.remove()
    toks = []
    for s in strings:
        toks.append([int(t[1:]) for t in s.split() if t.startswith("t")])
    return strings, toks, logits_log


def maxdiff(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max()), float((a - b).abs().max() / (b.abs().max() + 1e-30))


def gen_case(args, model, sd, lmp, prefix, B, t0):
    # ------------------------------------------------------------------ case "gen": greedy tokens + logits
    with torch.no_grad():
        strings, toks, logits_log = ref_generate_tokens(model, prefix, N_STEPS, stop_id=0)
        strings2, toks2, _ = ref_generate_tokens(model, prefix, N_STEPS, stop_id=0, top_p=0.1, temperature=0.3)
    toks = np.asarray(toks, dtype=np.int64)
    assert np.array_equal(toks, np.asarray(toks2)), "sampling params changed greedy result (SURVEY A16)"
    L = torch.stack(logits_log)                                             # (steps,B,V)
    top2 = torch.topk(L, 2, dim=-1).values
    gaps = (top2[..., 0] - top2[..., 1]).numpy()
    print(f"reference tokens ({time.time() - t0:.1f}s):\n{toks}\nmin top-2 gap {gaps.min():.4f}")
    rec = {}
    with torch.no_grad():
        otoks = O.generate_batch(sd, lmp, prefix, N_STEPS, 0.8, 1.0, 0, record=rec).numpy()
    assert np.array_equal(otoks, toks), (otoks, toks)
    d = maxdiff(torch.stack(rec["logits"]), L)
    print(f"oracle logits vs reference: {d[0]:.3e} abs")
    assert d[0] < 2e-3

    # EOS semantics: stop id := token row 0 produced at step 3 -> B=1 run must break right after it
    stop = int(toks[0, 3])
    with torch.no_grad():
        s_eos, t_eos, _ = ref_generate_tokens(model, prefix[:1], N_STEPS, stop_id=stop)
        s_eos2, t_eos2, _ = ref_generate_tokens(model, prefix, N_STEPS, stop_id=stop)
    print("eos case:", stop, t_eos, [len(t) for t in t_eos2])
    np.savez_compressed(
        os.path.join(args.out, "gen.npz"),
        seed=SEED, B=B, steps=N_STEPS, tokens=toks, logits_sub=L[:, :, SUB_VOCAB].numpy(), sub_vocab=SUB_VOCAB,
        logits_step0=L[0].numpy(), top2_gap=gaps,
        eos_stop_id=stop, eos_b1_tokens=np.asarray(t_eos[0], dtype=np.int64),
        eos_b2_len=np.asarray([len(t) for t in t_eos2], dtype=np.int64),
        eos_b2_row0=np.asarray(t_eos2[0], dtype=np.int64), eos_b2_row1=np.asarray(t_eos2[1], dtype=np.int64),
    )



def long30_case(args, model, sd):
    # ------------------------------------------------------------------ case "long30": 30 s -> 7 crops
    if True:
        a30 = torch.from_numpy(synth.make_clip(900, 960000))[None]
        with torch.no_grad():
            proj_ref, _, od = model.audio_encoder(a30)
            ot = {}
            proj_or = O.audio_encoder(sd, a30, ot)
        d = maxdiff(proj_or, proj_ref)
        print(f"long30: crops={ot.get('n_crops')} oracle vs reference projected {d[0]:.3e}")
        assert ot["n_crops"] == 7 and d[1] < 1e-4
        fw = od["framewise_output"]
        np.savez_compressed(
            os.path.join(args.out, "long30.npz"),
            clip_idx=900, n_samples=960000, n_crops=7,
            latent=od["latent_output"].numpy(), framewise32=fw[:, 0::32, :].numpy(),
            audio_ds=O.downsample(proj_ref).numpy(),
        )


LATE_STEPS = 300
LATE_KEEP = (63, 150, 299)          # step index i <-> context T = 389 + i keys (452, 539, 688)


def late_case(args, model, prefix, t0):
    """Late positions: the reference's own loop for 300 steps at B=2 (contexts 389..688, beyond one
    448-key attention chunk of the HIP decode kernel and at BASELINE config 3's max_len).  Stored: every greedy token, the
    top-2 logit gap of every step, and sub-sampled last-position logits of steps 63 / 150 / 299."""
    with torch.no_grad():
        _, toks, logits_log = ref_generate_tokens(model, prefix, LATE_STEPS, stop_id=-1)
    toks = np.asarray(toks, dtype=np.int64)
    assert toks.shape == (prefix.shape[0], LATE_STEPS), toks.shape
    gaps = np.stack([(lambda t2: (t2[..., 0] - t2[..., 1]).numpy())(torch.topk(l, 2, dim=-1).values) for l in logits_log])
    print(f"late: {LATE_STEPS} reference steps ({time.time() - t0:.1f}s); min top-2 gap {gaps.min():.4f} at step "
          f"{int(np.argmin(gaps.min(1)))}; tokens[:, -5:] = {toks[:, -5:].tolist()}")
    g12 = np.load(os.path.join(args.out, "gen.npz"))["tokens"]
    assert np.array_equal(toks[:, : g12.shape[1]], g12)
    np.savez_compressed(
        os.path.join(args.out, "late.npz"),
        seed=SEED, B=toks.shape[0], steps=LATE_STEPS, tokens=toks, top2_gap=gaps.astype(np.float32),
        keep_steps=np.asarray(LATE_KEEP), sub_vocab=SUB_VOCAB,
        logits_sub=torch.stack([logits_log[i][:, SUB_VOCAB] for i in LATE_KEEP]).numpy(),
        logits_max=torch.stack([logits_log[i].max(-1).values for i in LATE_KEEP]).numpy(),
    )


B32_STEPS = 64                      # BASELINE configs[1]: max_len = 64 -- the whole benchmarked run is reference-pinned
B32_LOGIT_STEPS = tuple(range(8)) + tuple(range(15, 64, 8))     # steps whose sub-vocabulary logits are stored


def b32_case(args, model, t0):
    """BASELINE configs[1]'s batch (the bench.py workload: examples 0..31) through the reference itself: all 64 greedy steps
    of all 32 rows (max_len = 64), a sub-sample of every row's prefix, every row's maximum logit and top-2 gap at every step,
    and the last-position logits (sub-vocabulary) of steps 0..7, 15, 23, ..., 63."""
    a1, a2, ids = synth.make_batch(32)
    with torch.no_grad():
        prefix, _, _ = model.generate_prefix_inference({"audio1": torch.from_numpy(a1), "audio2": torch.from_numpy(a2),
                                                        "input": {"input_ids": torch.from_numpy(ids)}})
        print(f"b32 prefix ({time.time() - t0:.1f}s)")
        _, toks, logits_log = ref_generate_tokens(model, prefix, B32_STEPS, stop_id=-1)
    toks = np.asarray(toks, dtype=np.int64)
    L = torch.stack(logits_log)                                             # (steps,32,V)
    top2 = torch.topk(L, 2, dim=-1).values
    gaps = (top2[..., 0] - top2[..., 1]).numpy()
    print(f"b32 tokens ({time.time() - t0:.1f}s): min top-2 gap {gaps.min():.4f}\n{toks}")
    ls = np.asarray(B32_LOGIT_STEPS)
    np.savez_compressed(os.path.join(args.out, "b32.npz"), steps=B32_STEPS, tokens=toks, top2_gap=gaps,
                        prefix_sub=prefix[:, ::7, ::5].numpy(), logit_steps=ls, logits_sub=L[ls][:, :, SUB_VOCAB].numpy(),
                        sub_vocab=SUB_VOCAB, logits_max=L.max(-1).values.numpy())


B64_STEPS = 16
B32LONG_STEPS = 300
B32LONG_KEEP = (63, 150, 299)


def b32long_case(args, model, t0):
    """BASELINE configs[2]'s per-rank workload through the reference itself: the 32 benchmarked examples for max_len = 300 (the
    reference re-forwards the whole 389..688-token sequence per step: ~1.5 h on the build container's 8 cores).  Stored: all
    32 x 300 greedy tokens, every step's maximum logit and top-2 gap, sub-vocabulary logits of steps 63 / 150 / 299."""
    a1, a2, ids = synth.make_batch(32)
    with torch.no_grad():
        prefix, _, _ = model.generate_prefix_inference({"audio1": torch.from_numpy(a1), "audio2": torch.from_numpy(a2),
                                                        "input": {"input_ids": torch.from_numpy(ids)}})
        _, toks, logits_log = ref_generate_tokens(model, prefix, B32LONG_STEPS, stop_id=-1)
    toks = np.asarray(toks, dtype=np.int64)
    assert toks.shape == (32, B32LONG_STEPS)
    gaps = np.stack([(lambda t2: (t2[..., 0] - t2[..., 1]).numpy())(torch.topk(l, 2, dim=-1).values) for l in logits_log])
    print(f"b32long: {B32LONG_STEPS} reference steps of 32 rows ({time.time() - t0:.1f}s); min top-2 gap {gaps.min():.4f} at step "
          f"{int(np.argmin(gaps.min(1)))}")
    b32 = np.load(os.path.join(HERE, "b32.npz"))["tokens"]
    assert np.array_equal(toks[:, : b32.shape[1]], b32)
    np.savez_compressed(os.path.join(args.out, "b32long.npz"), steps=B32LONG_STEPS, tokens=toks, top2_gap=gaps.astype(np.float32),
                        keep_steps=np.asarray(B32LONG_KEEP), sub_vocab=SUB_VOCAB,
                        logits_sub=torch.stack([logits_log[i][:, SUB_VOCAB] for i in B32LONG_KEEP]).numpy(),
                        logits_max=torch.stack([l.max(-1).values for l in logits_log]).numpy())


def cfg3_case(args, model, t0):
    """BASELINE configs[3]'s shape through the REFERENCE itself: 2 examples of 2 x 30 s clips (960,000 samples -> 7 encoder
    crops per clip, htsat.py:908-936), `generate_prefix_inference` + 16 steps of the unmodified `_generate_batch`."""
    a1, a2, ids = synth.make_batch(CFG3_ROWS, n_samples=30 * spec.SAMPLE_RATE)
    with torch.no_grad():
        prefix, _, _ = model.generate_prefix_inference({"audio1": torch.from_numpy(a1), "audio2": torch.from_numpy(a2),
                                                        "input": {"input_ids": torch.from_numpy(ids)}})
        _, toks, logits_log = ref_generate_tokens(model, prefix, CFG3_STEPS, stop_id=-1)
    toks = np.asarray(toks, dtype=np.int64)
    L = torch.stack(logits_log)
    top2 = torch.topk(L, 2, dim=-1).values
    gaps = (top2[..., 0] - top2[..., 1]).numpy()
    print(f"cfg3 (2 x 30 s) tokens ({time.time() - t0:.1f}s): min top-2 gap {gaps.min():.4f}\n{toks}")
    np.savez_compressed(os.path.join(args.out, "cfg3.npz"), rows=CFG3_ROWS, steps=CFG3_STEPS, n_samples=30 * spec.SAMPLE_RATE,
                        tokens=toks, top2_gap=gaps, prefix_sub=prefix[:, ::3, ::5].numpy(), sub_vocab=SUB_VOCAB,
                        logits_sub=L[:, :, SUB_VOCAB].numpy(), logits_max=L.max(-1).values.numpy())


CFG3_ROWS = 2
CFG3_STEPS = 16


def b64_tail_case(args, model, t0):
    """examples 32..63 (the second half of the north_star's 64-example batch) through the reference: 16 greedy steps + prefix"""
    a1, a2, ids = synth.make_batch(32, first=32)
    with torch.no_grad():
        prefix, _, _ = model.generate_prefix_inference({"audio1": torch.from_numpy(a1), "audio2": torch.from_numpy(a2),
                                                        "input": {"input_ids": torch.from_numpy(ids)}})
        _, toks, logits_log = ref_generate_tokens(model, prefix, B64_STEPS, stop_id=-1)
    toks = np.asarray(toks, dtype=np.int64)
    L = torch.stack(logits_log)
    top2 = torch.topk(L, 2, dim=-1).values
    print(f"b64 tail tokens ({time.time() - t0:.1f}s): min top-2 gap {float((top2[..., 0] - top2[..., 1]).min()):.4f}")
    np.savez_compressed(os.path.join(args.out, "b64tail.npz"), first=32, steps=B64_STEPS, tokens=toks,
                        prefix_sub=prefix[:, ::13, ::9].numpy(), logits_max=L.max(-1).values.numpy())


FWD_ANSWER_LEN = 12
FWD_FROM = 380


def forward_case(args, model, a1t, a2t, idst, prefix, t0):
    """The training-time forward (mellow.py:89-98 -> decoder.py:57-90): `model(input_dict).logits` over the sequence
    [prefix | embed(answer)] -- every position's logits, kept for positions >= FWD_FROM on the sub-vocabulary."""
    g = torch.Generator().manual_seed(SEED + 77)
    ans = torch.randint(0, 49152, (idst.shape[0], FWD_ANSWER_LEN), generator=g)
    with torch.no_grad():
        out = model({"audio1": a1t, "audio2": a2t, "input": {"input_ids": idst}, "answer": {"input_ids": ans}})
        emb = model.caption_decoder.lm.model.embed_tokens(ans)
        direct = model.caption_decoder.lm(inputs_embeds=torch.cat((prefix, emb), 1)).logits
    logits = out.logits
    assert logits.shape == (idst.shape[0], prefix.shape[1] + FWD_ANSWER_LEN, 49152)
    assert torch.equal(logits, direct)
    tail = logits[:, FWD_FROM:]
    print(f"forward: logits {tuple(logits.shape)}, |max| {float(tail.abs().max()):.3f} ({time.time() - t0:.1f}s)")
    np.savez_compressed(os.path.join(args.out, "forward.npz"), answer_ids=ans.numpy(), from_pos=FWD_FROM, sub_vocab=SUB_VOCAB,
                        logits_sub=tail[:, :, SUB_VOCAB].numpy(), logits_max=tail.max(-1).values.numpy(),
                        argmax=tail.argmax(-1).numpy(), answer_embed_sub=emb[:, :, ::9].numpy())


def _examples(idx):
    return synth.make_examples(idx)


RAGGED_EXAMPLES = (0, 1, 2)
EOS_EXAMPLES = (1, 2, 4, 3)         # rows that emit EOS_STOP at steps 8 / 17 / 3 / never (within 20 steps)
EOS_STOP = 42274
EOS_MAXLEN = 24


def ragged_eos_cases(args, model, sd, lmp, t0, want):
    """B=3 (not a multiple of anything) greedy tokens, and the reference's stop rule with rows that hit the stop id at
    different steps (wrapper.py:241-254): the loop ends after the first step at which every row has produced it."""
    idx = sorted(set(RAGGED_EXAMPLES) | set(EOS_EXAMPLES))
    a1, a2, ids = _examples(idx)
    with torch.no_grad():
        prefix, _, _ = model.generate_prefix_inference({"audio1": torch.from_numpy(a1), "audio2": torch.from_numpy(a2),
                                                        "input": {"input_ids": torch.from_numpy(ids)}})
    pos = {e: i for i, e in enumerate(idx)}
    if want("ragged"):
        p3 = prefix[[pos[e] for e in RAGGED_EXAMPLES]]
        with torch.no_grad():
            _, toks, logits_log = ref_generate_tokens(model, p3, N_STEPS, stop_id=-1)
        toks = np.asarray(toks, dtype=np.int64)
        print(f"ragged B=3 ({time.time() - t0:.1f}s):\n{toks}")
        np.savez_compressed(os.path.join(args.out, "ragged3.npz"), examples=np.asarray(RAGGED_EXAMPLES), steps=N_STEPS,
                            tokens=toks, prefix_row2=p3[2].numpy(),
                            logits_sub=torch.stack(logits_log)[:, :, SUB_VOCAB].numpy(), sub_vocab=SUB_VOCAB)
    if want("eos"):
        out = {}
        for name, rows in (("all_stop", EOS_EXAMPLES[:3]), ("one_never", EOS_EXAMPLES)):
            p = prefix[[pos[e] for e in rows]]
            with torch.no_grad():
                strings, toks, logits_log = ref_generate_tokens(model, p, EOS_MAXLEN, stop_id=EOS_STOP)
                _, free, _ = ref_generate_tokens(model, p, EOS_MAXLEN, stop_id=-1)
            n_steps = len(logits_log)
            lens = [len(t) for t in toks]
            print(f"eos {name}: rows {rows} steps run {n_steps} lengths {lens} ({time.time() - t0:.1f}s)")
            out[f"{name}_examples"] = np.asarray(rows)
            out[f"{name}_steps"] = n_steps
            out[f"{name}_len"] = np.asarray(lens, dtype=np.int64)
            out[f"{name}_free_tokens"] = np.asarray(free, dtype=np.int64)          # the same rows with no stop id
            for r, t in enumerate(toks):
                out[f"{name}_row{r}"] = np.asarray(t, dtype=np.int64)
        assert out["all_stop_steps"] < EOS_MAXLEN and len(set(out["all_stop_len"].tolist())) == 3
        np.savez_compressed(os.path.join(args.out, "eos_mixed.npz"), stop_id=EOS_STOP, max_len=EOS_MAXLEN, **out)


EX_SEED = 20250
EX_STEPS = 300
EX_PROMPT = "what is the primary sound event present in the clip? a) dog barking b) chirping birds c) car engine d) clapping"


def example_ids(prompt, L=129):
    """Tokenizer stand-in (the SmolLM2 files are not available offline): deterministic ids per word, right-padded with id 1 to
    129 -- the same function the GPU test's stub tokenizer uses, so the prompt ids of both sides are equal by construction."""
    ids = [17 + (sum(w.encode()) * 7919 + i * 104729) % 49000 for i, w in enumerate(prompt.split())][:L]
    return ids + [1] * (L - len(ids))


def example_case(args, model, t0):
    """BASELINE configs[0] / reference example.py:20-31: the two fixture clips + the README prompt through the REFERENCE's own
    `preprocess_audio` (wrapper.py:141-179: decode, resample to 32 kHz, flatten, tile 1.wav / crop 2.wav at a `random` offset),
    `generate_prefix_inference` and the unmodified `_generate_batch` loop for max_len = 300.  Stored: the decoded int16 PCM of
    both fixtures (data), the seed and the crop offset it produced, checks of the two preprocessed arrays, the prompt ids and
    all 300 greedy tokens.  The resampler inside is this build's restatement (torchaudio absent: parity unpinned there)."""
    import random
    import wave
    import yaml
    from argparse import Namespace
    from mellow.wrapper import MellowWrapper
    pcm, srs = [], []
    for name in ("1.wav", "2.wav"):
        with wave.open(os.path.join(REF, "resource", name), "rb") as w:
            assert w.getnchannels() == 1 and w.getsampwidth() == 2
            srs.append(w.getframerate())
            pcm.append(np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").copy())
    w = MellowWrapper.__new__(MellowWrapper)
    with open(os.path.join(REF, "mellow", "config", "v0.yaml")) as f:
        w.args = Namespace(**yaml.safe_load(f))
    w.use_cuda, w.device = False, "cpu"
    random.seed(EX_SEED)
    a1 = w.preprocess_audio([os.path.join(REF, "resource", "1.wav")], resample=True).squeeze(1)      # wrapper.py:277-278
    a2 = w.preprocess_audio([os.path.join(REF, "resource", "2.wav")], resample=True).squeeze(1)
    assert a1.shape == (1, 320000) and a2.shape == (1, 320000) and a1.dtype == torch.float32
    # recover the crop offset the reference drew: replay the draw (2.wav resamples to 323,585 > 320,000 samples)
    random.seed(EX_SEED)
    n2 = int(np.ceil(320 * len(pcm[1]) / 441))
    start = random.randrange(n2 - 320000)
    ids = torch.tensor([example_ids(EX_PROMPT)], dtype=torch.int64)
    with torch.no_grad():
        prefix, _, _ = model.generate_prefix_inference({"audio1": a1, "audio2": a2, "input": {"input_ids": ids}})
        _, toks, logits_log = ref_generate_tokens(model, prefix, EX_STEPS, stop_id=-1)
    toks = np.asarray(toks, dtype=np.int64)
    gaps = np.stack([(lambda t2: (t2[..., 0] - t2[..., 1]).numpy())(torch.topk(l, 2, dim=-1).values) for l in logits_log])
    print(f"example: n1 {len(pcm[0])} -> {int(np.ceil(320 * len(pcm[0]) / 441))} (tiled), n2 {len(pcm[1])} -> {n2} (crop at {start}); "
          f"{EX_STEPS} reference steps ({time.time() - t0:.1f}s); min top-2 gap {gaps.min():.4f}; tokens[:8] {toks[0, :8].tolist()}")
    np.savez_compressed(
        os.path.join(args.out, "example.npz"),
        pcm1=pcm[0], pcm2=pcm[1], sr1=srs[0], sr2=srs[1], seed=EX_SEED, crop_start2=start, prompt=EX_PROMPT,
        input_ids=ids.numpy(), steps=EX_STEPS, tokens=toks, top2_gap=gaps.astype(np.float32),
        audio1_sub=a1[0, ::61].numpy(), audio2_sub=a2[0, ::61].numpy(),
        audio1_sum=np.float64(a1.double().sum()), audio2_sum=np.float64(a2.double().sum()),
        audio1_abs=np.float64(a1.double().abs().sum()), audio2_abs=np.float64(a2.double().abs().sum()),
        prefix_sub=prefix[:, ::7, ::5].numpy(),
    )


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=HERE)
    ap.add_argument("--skip-long", action="store_true")
    ap.add_argument("--only", default="", help="comma list of cases to (re)generate: enc10,gen,long30,late,ragged,eos,forward,b32,b64,cfg3,b32long,example "
                                               "(default: all); enc10 is always computed (the others start from its prefix)")
    args = ap.parse_args()
    only = set(filter(None, args.only.split(",")))

    def want(case):
        return not only or case in only
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    t0 = time.time()
    sd = synth.make_state_dict(SEED)
    print(f"synthetic checkpoint: {len(sd)} entries ({time.time() - t0:.1f}s)")
    model, cfg = build_reference(sd)
    lmp = O.LMParams()

    # ------------------------------------------------------------------ case "enc10": B=2, 10 s
    B = 2
    a1, a2, ids = synth.make_batch(B)
    a1t, a2t, idst = torch.from_numpy(a1), torch.from_numpy(a2), torch.from_numpy(ids)
    taps = {}
    hooks = []
    ht = model.audio_encoder.base.htsat

    def tap(name):
        def f(m, i, o):
            if name not in taps:         # first call = audio1 pass (mellow.py:105)
                taps[name] = (o[0] if isinstance(o, tuple) else o).detach().clone()
        return f

    hooks.append(ht.spectrogram_extractor.register_forward_hook(tap("power")))
    hooks.append(ht.logmel_extractor.register_forward_hook(tap("logmel")))
    hooks.append(ht.bn0.register_forward_hook(tap("bn0_raw")))           # (B,64,1001,1) transposed view
    hooks.append(ht.patch_embed.register_forward_hook(tap("patch")))
    for s in range(4):
        hooks.append(ht.layers[s].register_forward_hook(tap(f"stage{s}")))
    hooks.append(model.audio_encoder.projection.register_forward_hook(tap("projected")))
    with torch.no_grad():
        prefix, od1, od2 = model.generate_prefix_inference({"audio1": a1t, "audio2": a2t, "input": {"input_ids": idst}})
    for h in hooks:
        h.remove()
    print(f"reference prefix: {tuple(prefix.shape)} ({time.time() - t0:.1f}s)")

    # oracle on the same inputs
    otaps = {}
    with torch.no_grad():
        oprefix = O.generate_prefix_inference(sd, a1t, a2t, idst, otaps)
    ref_logmel_bn = taps["bn0_raw"].transpose(1, 3)
    checks = {
        "power": (otaps["power"], taps["power"]),
        "logmel": (otaps["logmel"], taps["logmel"]),
        "logmel_bn": (otaps["logmel_bn"], ref_logmel_bn),
        "patch": (otaps["patch"], taps["patch"]),
        **{f"stage{s}": (otaps[f"stage{s}"], taps[f"stage{s}"]) for s in range(4)},
        "latent": (otaps["latent"], od1["latent_output"]),
        "framewise": (otaps["framewise"], od1["framewise_output"]),
        "embedding": (otaps["embedding"], od1["embedding"]),
        "projected": (otaps["projected"], taps["projected"]),
        "prefix": (oprefix, prefix),
    }
    print("oracle vs imported reference (max abs, max abs / max|ref|):")
    for k, (o, r) in checks.items():
        d = maxdiff(o, r)
        print(f"  {k:10s} {d[0]:.3e} {d[1]:.3e}   shape {tuple(r.shape)}")
        assert d[1] < 1e-4, (k, d)

    tok_idx = np.arange(0, 4096, 41)       # 100 patch tokens
    fw = od1["framewise_output"]
    assert torch.equal(fw[:, 0::32, :].repeat_interleave(32, dim=1), fw)   # only 32 distinct rows
    if want("enc10"):
      np.savez_compressed(
        os.path.join(args.out, "enc10.npz"),
        seed=SEED, B=B,
        power_sub=taps["power"][:, 0, ::50, ::8].numpy(),                  # (B,21,65)
        logmel=taps["logmel"][:, 0].numpy().astype(np.float32),           # (B,1001,64)
        logmel_bn_sub=ref_logmel_bn[:, 0, ::10, :].numpy(),               # (B,101,64)
        patch_sub=taps["patch"][:, tok_idx, :].numpy(), tok_idx=tok_idx,
        stage0_sub=taps["stage0"][:, ::11, :].numpy(),
        stage1_sub=taps["stage1"][:, ::5, :].numpy(),
        stage2=taps["stage2"].numpy(),
        stage3=taps["stage3"].numpy(),
        latent=od1["latent_output"].numpy(),
        framewise32=fw[:, 0::32, :].numpy(),                                # (B,32,527)
        embedding33=torch.cat((od1["embedding"][:, :1], od1["embedding"][:, 1::32]), 1).numpy(),
        projected33=torch.cat((taps["projected"][:, :1], taps["projected"][:, 1::32]), 1).numpy(),
        prefix=prefix.numpy(),                                              # (B,389,576)
    )

    if want("gen"):
        gen_case(args, model, sd, lmp, prefix, B, t0)
    if want("long30") and not args.skip_long:
        long30_case(args, model, sd)
    if want("late"):
        late_case(args, model, prefix, t0)
    if want("forward"):
        forward_case(args, model, a1t, a2t, idst, prefix, t0)
    if "b32" in only or (not only and not args.skip_long):
        b32_case(args, model, t0)
    if "b64" in only or (not only and not args.skip_long):
        b64_tail_case(args, model, t0)
    if "cfg3" in only or (not only and not args.skip_long):
        cfg3_case(args, model, t0)
    if "b32long" in only:               # only on request: 1.5 hours
        b32long_case(args, model, t0)
    if want("ragged") or want("eos"):
        ragged_eos_cases(args, model, sd, lmp, t0, want)
    if want("example"):
        example_case(args, model, t0)
    print(f"done ({time.time() - t0:.1f}s)")


if __name__ == "__main__":
    main()
