"""`MellowWrapper` — drop-in for the reference's public API (reference mellow/wrapper.py:25-287,
re-exported by mellow/__init__.py:1):

    from mellow_amd import MellowWrapper
    mellow = MellowWrapper(config="v0", model="v0", device=0, use_cuda=True)
    response = mellow.generate(examples=[[path1, path2, prompt]], max_len=300, top_p=0.8, temperature=1.0)

Host code stays Python (yaml config, checkpoint loading, wav ingest, tokenisation); everything from the
(B, 320000) waveforms to the generated token ids runs in libmellow_hip.so on the MI355X through the C ABI of
include/mellow_hip.h.  There is no CPU model path: `use_cuda=False` / `device="cpu"` raise.

What is kept from the reference, quirks included: class attributes `model_repo`/`model_name`;
`ValueError` for an unknown model; `config/<config>.yaml` key layout; strict `state_dict` load with the
'module.' retry; prompt right-padded with '!' to 129 ids; sep = token 0; pads attended; multi-channel wav
flattened not mixed; crop start from the unseeded `random` module; sampling parameters accepted but the result
is greedy for every value (the reference's top-p filter never removes the arg-max, wrapper.py:220-232); the
loop stops only when every row has produced the stop id; text is cut at the first '<|endoftext|>'.
Deviations: `tqdm` progress output is not produced; the B>1/steps==1 mis-shape and B==1/steps==1 crash of
reference wrapper.py:251-253 are not reproduced (one string per example is always returned).
"""
from __future__ import annotations

import argparse
import math
import os
import warnings
from collections import OrderedDict
from pathlib import Path
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import yaml

from . import spec
from .audio import load_audio_into_tensor
from .engine import DEFAULT_PRECISION, Engine, EngineError
from .spec import LMConfig


def get_model_class(model_type: str):
    """reference mellow/model/model.py:3-7."""
    if model_type == "Mellow":
        return Engine
    raise NotImplementedError


def get_audio_encoder(name: str):
    """reference mellow/model/audio.py:3-7."""
    if name == "HTSAT":
        return "HTSAT"
    raise Exception("The audio encoder name {} is incorrect or not supported".format(name))


class MellowWrapper:
    """A class for interfacing the Mellow model on MI355X."""

    model_repo = "soham97/mellow"
    model_name = {"v0": "v0.ckpt", "v0_s": "v0_s.ckpt"}

    def __init__(self, config, model, device, use_cuda=True, *, checkpoint: Optional[str] = None,
                 state_dict: Optional[Dict[str, torch.Tensor]] = None, tokenizer=None, max_positions: Optional[int] = None,
                 data_parallel: Optional[bool] = None, precision: Optional[str] = None,
                 engine_options: Optional[Dict[str, int]] = None):
        """Reference signature `MellowWrapper(config, model, device, use_cuda=True)` (wrapper.py:35) plus keyword-only
        extensions: `checkpoint` (a local .ckpt instead of the hub download), `state_dict` (already loaded), `tokenizer`
        (an object with encode / encode_plus / decode), `max_positions` (prefix 389 + max_len may not exceed it; default: the LM's
        max_position_embeddings, 8192), `data_parallel` (True or MELLOW_DATA_PARALLEL=1: shard `generate` over the ranks of an
        initialised torch.distributed group, one process per GPU, every rank calling with the SAME examples -- checked; default
        off: like the reference, every process answers its own examples), `precision` ("f32x3" (default) | "f32" | "fp8"),
        `engine_options` (named options of the HIP library, mellow_engine_set_option; default none: the library's defaults --
        the library itself reads no environment variable)."""
        self.supported_versions = self.model_name.keys()
        if model not in self.supported_versions:
            raise ValueError(f"The model {model} is not supported. The supported versions are {str(self.supported_versions)}")
        self.parent_path = Path(os.path.realpath(__file__)).parent
        self.config_path = os.path.join(self.parent_path, "config", config + ".yaml")
        self.use_cuda = use_cuda
        self.device = device
        self._state_dict = state_dict
        self.model_path = self._resolve_checkpoint(model, checkpoint) if state_dict is None else "<state_dict>"
        self._tokenizer_override = tokenizer
        self._max_positions = max_positions
        self._data_parallel = data_parallel
        # numeric mode of the dense GEMMs (include/mellow_hip.h): "f32x3" fp32-accurate bf16-split (default = the mode
        # bench.py reports), "f32" exact fp32 MFMA, "fp8" BASELINE config 5; keyword or MELLOW_PRECISION
        self._precision = precision or os.environ.get("MELLOW_PRECISION") or DEFAULT_PRECISION
        self._engine_options = dict(engine_options or {})
        self.model, self.tokenizer, self.args = self.get_model_and_tokenizer(config_path=self.config_path)

    # ---- construction -------------------------------------------------------------------------------------
    def _resolve_checkpoint(self, model: str, checkpoint: Optional[str]) -> str:
        if checkpoint is not None:
            return checkpoint
        name = self.model_name[model]
        local = os.environ.get("MELLOW_CKPT_DIR")
        if local and os.path.exists(os.path.join(local, name)):
            return os.path.join(local, name)
        try:  # reference wrapper.py:41-42
            from huggingface_hub.file_download import hf_hub_download
            path = hf_hub_download(self.model_repo, name)
            try:
                hf_hub_download(self.model_repo, "config.json")   # the reference's download counter
            except Exception:
                pass
            return path
        except Exception as e:
            raise FileNotFoundError(
                f"checkpoint {name} not available offline: pass checkpoint=..., state_dict=..., or set MELLOW_CKPT_DIR ({e})")

    def read_config_as_args(self, config_path):
        """yaml -> argparse.Namespace whose fields are plain dicts (reference wrapper.py:51-57)."""
        with open(config_path, "r") as f:
            yml_config = yaml.load(f, Loader=yaml.FullLoader)
        return argparse.Namespace(**{k: v for k, v in yml_config.items()})

    def get_model_and_tokenizer(self, config_path):
        args = self.read_config_as_args(config_path)
        args.model["decoder"]["prefix_dim"] = args.model["encoder"]["d_proj"]
        get_model_class(model_type=args.model["model_type"])                  # NotImplementedError on unknown type
        get_audio_encoder(args.model["encoder"]["audioenc_name"])             # Exception on unknown encoder
        text_decoder = args.model["decoder"]["text_decoder"]
        if "smollm2" not in text_decoder.lower():                            # reference decoder.py:30-31
            raise ValueError(f"text decoder {text_decoder.lower()} not supported")
        if args.model["decoder"]["prefix_length"] != spec.PREFIX_LEN or args.data["text_tokenization_len"] != spec.TEXT_LEN \
                or args.model["encoder"]["d_proj"] != spec.D_PROJ or args.data["sampling_rate"] != spec.SAMPLE_RATE:
            raise ValueError("config does not describe the v0 geometry this engine is built for")
        if not self.use_cuda or isinstance(self.device, str):
            raise RuntimeError("MellowWrapper (MI355X engine) has no CPU path: pass use_cuda=True and an integer device")
        lm = LMConfig.load()
        engine = Engine(lm=lm, device=int(self.device), max_positions=self._max_positions, precision=self._precision,
                        options=self._engine_options)
        sd = self._state_dict
        if sd is None:
            sd = torch.load(self.model_path, map_location=torch.device("cpu"))
        params = 0
        for k, v in sd.items():
            kk = k[7:] if k.startswith("module.") else k
            if kk.endswith(("running_mean", "running_var", "num_batches_tracked", "attn_mask", "relative_position_index")) \
                    or kk == spec.LM + "lm_head.weight":
                continue
            params += math.prod(v.size())
        engine.load_state_dict(sd, strict=True)       # strict, with the 'module.' retry of wrapper.py:75-82 in the engine
        tokenizer = self._tokenizer_override
        if tokenizer is None:
            from transformers import AutoTokenizer
            tokenizer = AutoTokenizer.from_pretrained(text_decoder)
            tokenizer.add_special_tokens({"pad_token": "!"})
        model_path = self.model_path.split(os.path.sep)[-1]
        cfg_name = config_path.split(os.path.sep)[-1]
        print(f"model {model_path}, {cfg_name}, parameter count: {params}")
        return engine, tokenizer, args

    # ---- preprocessing ----------------------------------------------------------------------------------------
    def load_audio_into_tensor(self, audio_path, audio_duration, resample=True):
        return load_audio_into_tensor(audio_path, audio_duration, self.args.data["sampling_rate"], resample)

    def preprocess_audio(self, audio_files, resample):
        """-> float32 (B, segment_seconds*sampling_rate) on the engine's device (reference wrapper.py:170-179).
        With MELLOW_DEVICE_RESAMPLE=1 the resampling runs on the GPU (mellow_resample, the device twin of audio.resample):
        files are decoded on the host, resampled per file on the device, then tiled / cropped as in the reference."""
        if resample and os.environ.get("MELLOW_DEVICE_RESAMPLE") == "1":
            from .audio import fit_duration, load_wav
            sr_t = self.args.data["sampling_rate"]
            rows = []
            for f in audio_files:
                wav, sr = load_wav(str(f))
                w = self.model.resample(wav, sr, sr_t) if sr != sr_t else wav.to(self.model.tdev)
                rows.append(fit_duration(w.reshape(-1), self.args.data["segment_seconds"] * sr_t).to(torch.float32).reshape(1, -1))
            return torch.cat(rows, 0).to(self.model.tdev)
        # files are independent: decode / resample / tile-or-crop on a thread pool (torch releases the GIL in the conv1d of
        # the resampler); output order = input order.  The reference does this serially; its crop start is an unseeded
        # `random` draw per file (wrapper.py:164), so the draw order carries no meaning.
        def one(f):
            return self.load_audio_into_tensor(f, self.args.data["segment_seconds"], resample).reshape(1, -1)
        if len(audio_files) > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(len(audio_files), os.cpu_count() or 1, 32)) as pool:
                tensors = list(pool.map(one, audio_files))
        else:
            tensors = [one(f) for f in audio_files]
        return torch.cat(tensors, 0).to(self.model.tdev)

    def _encode_padded(self, text, L):
        """One prompt -> BatchEncoding padded / truncated to L ids.  The reference calls `tokenizer.encode_plus(...,
        pad_to_max_length=True)` (wrapper.py:186-190, transformers 4.46); newer transformers spell the padding
        `padding="max_length"`, and transformers >= 5 dropped `encode_plus` in favour of `tokenizer(...)` (same arguments)."""
        enc = getattr(self.tokenizer, "encode_plus", None) or self.tokenizer
        last = None
        for pad_kw in ({"padding": "max_length"}, {"pad_to_max_length": True}):
            try:
                return enc(text=text, add_special_tokens=True, truncation=True, max_length=L, return_tensors="pt", **pad_kw)
            except TypeError as e:      # this spelling is not known to the installed tokenizer
                last = e
        raise last

    def preprocess_text(self, prompts):
        """-> {'input_ids', 'attention_mask'} int64 (B, 129) (reference wrapper.py:181-195; the mask is never used)."""
        L = self.args.data["text_tokenization_len"]
        ids, masks = [], []
        for ttext in prompts:
            ttext = ttext + " <|endoftext|>" if "gpt" in self.args.model["decoder"]["text_decoder"] else ttext
            tok = self._encode_padded(ttext, L)
            ids.append(torch.as_tensor(tok["input_ids"]).reshape(-1))
            masks.append(torch.as_tensor(tok["attention_mask"]).reshape(-1))
        return {"input_ids": torch.stack(ids, 0), "attention_mask": torch.stack(masks, 0)}

    # ---- generation ---------------------------------------------------------------------------------------------
    def _dp(self):
        """(rank, world) of the data-parallel group `generate` shards over, (0, 1) when not distributed."""
        import torch.distributed as dist
        # opt-in (the reference has no such behaviour: under torchrun every rank normally holds its OWN examples)
        on = self._data_parallel is True or (self._data_parallel is None and os.environ.get("MELLOW_DATA_PARALLEL") == "1")
        if on and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return dist.get_rank(), dist.get_world_size()
        return 0, 1

    def _check_same_examples(self, examples):
        """Sharding is only meaningful when every rank was handed the same list: compare (count, content digest) across ranks and
        raise on every rank otherwise (a silent mismatch would return other ranks' texts, or hang in the gather).  The exchange
        runs over the process group's rendezvous store (mellow_amd.dist.agree_on_examples): the token all-gather is the only
        collective of the call."""
        from . import dist as mdist
        mdist.agree_on_examples(mdist.examples_signature(examples))

    def _clamp_max_len(self, entry_length: int) -> int:
        limit = self.model.max_new_tokens_limit()
        if entry_length > limit:
            # the reference treats max_len as a safety bound (the loop normally ends at the stop token, wrapper.py:247-249)
            warnings.warn(f"max_len {entry_length} exceeds what the engine's KV pages hold (prefix {spec.PREFIX_LEN} + "
                          f"max_len <= {limit + spec.PREFIX_LEN}); clamped to {limit}")
            return limit
        return entry_length

    def _generate_batch(self, audio1, audio2, input_ids, entry_length=300, top_p=0.8, temperature=1.0,
                        stop_token: str = "<|endoftext|>", n_total: Optional[int] = None):
        """Tokens for the rows given (this rank's shard under data parallelism), decoded for ALL `n_total` examples:
        the shards' token ids are all-gathered once (mellow_amd.dist, RCCL over xGMI under backend "nccl")."""
        stop_token_index = self.tokenizer.encode(stop_token)[0]
        entry_length = self._clamp_max_len(int(entry_length))
        rank, world = self._dp()
        n_local = int(audio1.shape[0])
        if n_local:
            toks, lens, steps, ftm = self.model.generate(audio1, audio2, input_ids, max_len=entry_length, top_p=top_p,
                                                         temperature=temperature, stop_id=stop_token_index)
            self.last_first_token_ms = ftm
        else:
            toks, lens = np.zeros((0, 0), dtype=np.int32), np.zeros((0,), dtype=np.int32)
        if world > 1:
            import torch.distributed as dist
            from . import dist as mdist
            dev = self.model.tdev if dist.get_backend() == "nccl" else torch.device("cpu")
            toks, lens = mdist.gather_tokens(toks, lens, int(n_total), entry_length, device=dev)
        # -1 = never computed: padding of shards that stopped earlier, or steps after a whole 32-row block had stopped
        rows = [r[r >= 0] for r in toks]
        return [self.tokenizer.decode(x).split("<|endoftext|>")[0] for x in rows]

    def generate(self, examples, max_len, top_p, temperature, stop_token="<|endoftext|>", audio_resample=True):
        r"""Produces text response for the given audio files and text prompts
        examples: (list<list>) each example is [audio path 1, audio path 2, text prompt]
        max_len: (int) maximum length for text generation
        top_p, temperature: accepted for API parity; decoding is greedy (see module docstring)
        stop_token: (str) token used to stop text generation
        audio_resample (bool) True for resampling audio. The model supports only 32 kHz

        With `data_parallel=True` (or MELLOW_DATA_PARALLEL=1) under an initialised torch.distributed group (one process per
        GPU, every rank calling with the same examples) the examples are sharded contiguously over the ranks, each rank ingests
        and runs only its shard, and every rank returns the full list."""
        audio_paths1, audio_paths2, text_prompts = [], [], []
        for example in examples:
            ap1, ap2, tp = example
            audio_paths1.append(ap1)
            audio_paths2.append(ap2)
            text_prompts.append(tp)
        rank, world = self._dp()
        n = len(examples)
        if n == 0:          # the reference fails in torch.cat(audio_tensors) (wrapper.py:178) on an empty list
            raise RuntimeError("torch.cat(): expected a non-empty list of Tensors")
        lo, hi = 0, n
        if world > 1:
            from .dist import shard_range
            self._check_same_examples(examples)
            lo, hi = shard_range(n, rank, world)
        if hi > lo:
            audio1 = self.preprocess_audio(audio_paths1[lo:hi], resample=audio_resample)
            audio2 = self.preprocess_audio(audio_paths2[lo:hi], resample=audio_resample)
            ids = self.preprocess_text(text_prompts[lo:hi])["input_ids"]
        else:
            audio1 = audio2 = torch.zeros((0, 1))
            ids = torch.zeros((0, spec.TEXT_LEN), dtype=torch.int64)
        return self._generate_batch(audio1, audio2, ids, entry_length=max_len, top_p=top_p,
                                    temperature=temperature, stop_token=stop_token, n_total=n)
