"""CPU: the tokeniser / detokeniser edge of the drop-in wrapper (reference wrapper.py:84-85, 181-195, 208, 251-254) with a REAL
byte-level BPE tokenizer object of the same family as SmolLM2's (GPT-2 style ByteLevel BPE, `<|endoftext|>` = id 0, pad token
'!'), trained locally on a few sentences because the SmolLM2 tokenizer files cannot be fetched offline (parity with the real
vocabulary stays unpinned; what is pinned here is every call the wrapper makes on the `transformers` tokenizer API)."""
import argparse

import numpy as np
import pytest
import torch

from mellow_amd import spec


@pytest.fixture(scope="module")
def bpe():
    tokenizers = pytest.importorskip("tokenizers")
    transformers = pytest.importorskip("transformers")
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers
    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    corpus = ["what is the difference between the two audios", "describe both sounds in detail!",
              "which one is louder? the first or the second", "a dog barks while rain falls"] * 50
    trainer = trainers.BpeTrainer(vocab_size=400, special_tokens=["<|endoftext|>"],
                                  initial_alphabet=pre_tokenizers.ByteLevel.alphabet())
    tok.train_from_iterator(corpus, trainer)
    t = transformers.PreTrainedTokenizerFast(tokenizer_object=tok, eos_token="<|endoftext|>", bos_token="<|endoftext|>",
                                             unk_token="<|endoftext|>")
    t.add_special_tokens({"pad_token": "!"})                       # reference wrapper.py:85
    return t


class _FakeEngine:
    """stands in for the HIP engine: returns a scripted token matrix (the GPU path is covered by tests -m gpu)"""

    def __init__(self, toks, lens, steps):
        self.toks, self.lens, self.steps = toks, lens, steps
        self.calls = []

    def max_new_tokens_limit(self):
        return 2048 - spec.PREFIX_LEN

    def generate(self, a1, a2, ids, max_len, top_p, temperature, stop_id):
        self.calls.append((tuple(ids.shape), max_len, stop_id))
        return self.toks[:, : self.steps], self.lens, self.steps, 1.0


def _wrapper(tokenizer, engine=None):
    from mellow_amd import MellowWrapper
    w = MellowWrapper.__new__(MellowWrapper)
    w.tokenizer = tokenizer
    w.model = engine
    w._data_parallel = False
    w.args = argparse.Namespace(data={"text_tokenization_len": spec.TEXT_LEN, "sampling_rate": 32000, "segment_seconds": 10},
                                model={"decoder": {"text_decoder": "HuggingFaceTB/SmolLM2-135M"}})
    return w


def test_preprocess_text_pads_with_bang_and_truncates(bpe):
    w = _wrapper(bpe)
    prompts = ["what is the difference between the two audios", "describe both " * 200, ""]
    out = w.preprocess_text(prompts)
    ids = out["input_ids"]
    assert ids.shape == (3, spec.TEXT_LEN) and ids.dtype == torch.int64 and out["attention_mask"].shape == ids.shape
    pad = bpe.encode("!")[0]
    assert pad == bpe.pad_token_id
    plain = bpe.encode(prompts[0])
    assert ids[0, : len(plain)].tolist() == plain and (ids[0, len(plain):] == pad).all()      # right-padded with '!'
    assert ids[1].tolist() == bpe.encode(prompts[1])[: spec.TEXT_LEN]                            # truncated, no pad
    assert (ids[2] == pad).all()                                                                # empty prompt: 129 pads
    assert int(ids.max()) < len(bpe) and int(ids.min()) >= 0
    assert bpe.encode("<|endoftext|>")[0] == 0                                                  # stop id = sep id = 0 (decoder.py:49)


def test_old_and_new_padding_spellings_are_both_accepted():
    """transformers 4.46 (the reference's pin) accepted `pad_to_max_length=True`; >= 4.5x wants `padding="max_length"`;
    >= 5 has no `encode_plus`.  The wrapper works with all three."""
    calls = []

    class Old:                                           # only knows the reference's spelling
        def encode_plus(self, text, add_special_tokens, truncation, max_length, pad_to_max_length, return_tensors):
            calls.append("old")
            return {"input_ids": torch.zeros(1, max_length, dtype=torch.int64), "attention_mask": torch.ones(1, max_length, dtype=torch.int64)}

    class New:                                           # no encode_plus at all: __call__ with padding=
        def __call__(self, text, add_special_tokens, truncation, max_length, padding, return_tensors):
            calls.append("new")
            return {"input_ids": torch.ones(1, max_length, dtype=torch.int64), "attention_mask": torch.ones(1, max_length, dtype=torch.int64)}

    assert _wrapper(Old()).preprocess_text(["x"])["input_ids"].shape == (1, spec.TEXT_LEN)
    assert _wrapper(New()).preprocess_text(["x", "y"])["input_ids"].shape == (2, spec.TEXT_LEN)
    assert calls == ["old", "new", "new"]


def test_detokenise_and_cut_at_first_stop(bpe):
    """`tokenizer.decode(row).split('<|endoftext|>')[0]` per row (wrapper.py:251-254): text before the first stop token; rows that
    never stop keep everything; -1 (never computed) is dropped before decoding; max_len beyond the KV pages is clamped."""
    a = bpe.encode("a dog barks")
    b = bpe.encode("rain falls while the second one is louder")
    L = max(len(a), len(b)) + 3
    toks = np.full((2, L), -1, dtype=np.int32)
    toks[0, : len(a)] = a
    toks[0, len(a)] = 0                                   # stop id, then garbage the reference would also discard
    toks[0, len(a) + 1: len(a) + 3] = bpe.encode("the")[0]
    toks[1, : len(b)] = b                                 # never stops; trailing -1 = steps its block never computed
    eng = _FakeEngine(toks, np.asarray([len(a), len(b)], dtype=np.int32), L)
    w = _wrapper(bpe, eng)
    ids = w.preprocess_text(["p1", "p2"])["input_ids"]
    out = w._generate_batch(torch.zeros(2, 8), torch.zeros(2, 8), ids, entry_length=L)
    assert out == ["a dog barks", "rain falls while the second one is louder"]
    assert eng.calls[-1] == ((2, spec.TEXT_LEN), L, 0)    # stop id passed to the engine = tokenizer.encode(stop_token)[0]
    with pytest.warns(UserWarning, match="clamped"):
        w._generate_batch(torch.zeros(2, 8), torch.zeros(2, 8), ids, entry_length=10000)
    assert eng.calls[-1][1] == 2048 - spec.PREFIX_LEN


def test_empty_example_list_fails_like_the_reference(bpe):
    """`generate([])`: the reference dies in `torch.cat(audio_tensors)` (wrapper.py:178); same exception type and message here."""
    w = _wrapper(bpe, _FakeEngine(np.zeros((0, 0), np.int32), np.zeros((0,), np.int32), 0))
    with pytest.raises(RuntimeError, match="non-empty list"):
        w.generate(examples=[], max_len=8, top_p=0.8, temperature=1.0)
    with pytest.raises(ValueError):                      # malformed example: unpacking error, like wrapper.py:272
        w.generate(examples=[["a.wav", "b.wav"]], max_len=8, top_p=0.8, temperature=1.0)
