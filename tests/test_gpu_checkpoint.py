"""GPU (-m gpu): the checkpoint-file path of the drop-in wrapper (reference wrapper.py:74-82: `torch.load(path,
map_location='cpu')`, `load_state_dict(strict)` with the 'module.' retry) and the loader's validation of all 479 keys.
The real v0.ckpt / v0_s.ckpt cannot be fetched offline: the files are `torch.save`s of the seeded synthetic state_dict,
which has the real key layout, shapes and dtypes."""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from mellow_amd import spec, synth

pytestmark = pytest.mark.gpu


class _IdTokenizer:
    def __init__(self, stop_id=0):
        self.stop_id = stop_id

    def encode(self, s):
        return [self.stop_id]

    def decode(self, ids):
        return " ".join("<|endoftext|>" if int(i) == self.stop_id else f"t{int(i)}" for i in np.atleast_1d(ids))


def _golden_strings(golden_dir, steps):
    g = np.load(os.path.join(golden_dir, "gen.npz"))
    return [" ".join(f"t{int(t)}" for t in row[:steps]) for row in g["tokens"]]


def _run(wrapper, steps=12):
    a1, a2, ids = synth.make_batch(2)
    return wrapper._generate_batch(torch.from_numpy(a1), torch.from_numpy(a2), torch.from_numpy(ids), entry_length=steps)


@pytest.mark.parametrize("prefix", ["", "module."])
def test_checkpoint_file_round_trip(synth_sd, tmp_path, golden_dir, prefix):
    """`MellowWrapper(checkpoint=path)`: a pickled state_dict, plain or DataParallel-prefixed (wrapper.py:78-82), gives
    the reference's golden tokens."""
    from mellow_amd import MellowWrapper
    path = tmp_path / "v0.ckpt"
    torch.save(OrderedDict((prefix + k, v) for k, v in synth_sd.items()), path)
    m = MellowWrapper(config="v0", model="v0", device=0, use_cuda=True, checkpoint=str(path), tokenizer=_IdTokenizer())
    assert m.model_path == str(path)
    assert [s.strip() for s in _run(m)] == _golden_strings(golden_dir, 12)
    m.model.close()


def test_model_name_resolves_checkpoint_dir(synth_sd, tmp_path, golden_dir, monkeypatch):
    """model="v0_s" -> v0_s.ckpt (class attribute `model_name`, wrapper.py:29-33) looked up in MELLOW_CKPT_DIR before the
    hub; v0 and v0_s share config "v0" (README.md:33-34)."""
    from mellow_amd import MellowWrapper
    torch.save(synth_sd, tmp_path / "v0_s.ckpt")
    monkeypatch.setenv("MELLOW_CKPT_DIR", str(tmp_path))
    m = MellowWrapper(config="v0", model="v0_s", device=0, use_cuda=True, tokenizer=_IdTokenizer())
    assert m.model_path.endswith("v0_s.ckpt")
    assert [s.strip() for s in _run(m, 4)] == _golden_strings(golden_dir, 4)
    m.model.close()
    monkeypatch.setenv("MELLOW_CKPT_DIR", str(tmp_path / "nowhere"))
    monkeypatch.setenv("HF_HUB_OFFLINE", "1")
    with pytest.raises(FileNotFoundError, match="v0.ckpt"):
        MellowWrapper(config="v0", model="v0", device=0, use_cuda=True, tokenizer=_IdTokenizer())


def _engine():
    from mellow_amd.engine import Engine
    return Engine(device=0)


def test_strict_load_failure_modes(synth_sd):
    """Missing key, unexpected key and wrong shape each fail and name the key (strict load, wrapper.py:76)."""
    from mellow_amd.engine import EngineError
    victim = spec.ENC + "layers.2.blocks.3.mlp.fc1.weight"
    sd = OrderedDict(synth_sd)
    del sd[victim]
    e = _engine()
    with pytest.raises(EngineError, match="missing key.*" + victim.replace(".", r"\.")):
        e.load_state_dict(sd)
    e.close()

    sd = OrderedDict(synth_sd)
    sd["audio_encoder.base.htsat.layers.0.blocks.0.attn.extra_table"] = torch.zeros(3)
    e = _engine()
    with pytest.raises(EngineError, match="unexpected key.*extra_table"):
        e.load_state_dict(sd)
    e.close()

    for key, shape in ((spec.LM + "model.layers.7.mlp.down_proj.weight", (576, 1535)),
                       (spec.ENC + "layers.1.blocks.1.attn_mask", (16, 64, 63)),
                       (spec.ENC + "bn0.running_var", (63,)),
                       (spec.ENC + "layers.3.blocks.0.attn.relative_position_bias_table", (225, 31))):
        sd = OrderedDict(synth_sd)
        sd[key] = torch.zeros(shape)
        e = _engine()
        with pytest.raises(EngineError, match="(size mismatch|rank mismatch).*" + key.split(".")[-1]):
            e.load_state_dict(sd)
        e.close()

    # ignored-but-legal entries of the real checkpoint: unused head, tied lm_head, BatchNorm counter
    sd = OrderedDict(synth_sd)
    for k in spec.UNUSED_KEYS + (spec.LM + "lm_head.weight",):
        assert k in sd
    e = _engine()
    e.load_state_dict(sd)
    assert len(e.required_keys()) + len(spec.UNUSED_KEYS) + 1 == len(sd) == 479
    e.close()


def test_index_and_mask_buffers_are_consumed_and_validated(synth_sd):
    """`relative_position_index` (int64) and `attn_mask` are persistent buffers of the reference's checkpoint
    (htsat.py:291, 412) that `load_state_dict` overwrites: the engine reads them from the checkpoint like the reference does
    (a different mask changes the output), rejects indices outside the 225-row bias table, and accepts int32 indices."""
    a1, _, _ = synth.make_batch(1)
    e = _engine()
    e.load_state_dict(synth_sd)
    base = e.encode(a1).cpu()
    e.close()

    key_i = spec.ENC + "layers.0.blocks.0.attn.relative_position_index"
    sd = OrderedDict(synth_sd)
    sd[key_i] = synth_sd[key_i].to(torch.int32)
    e = _engine()
    e.load_state_dict(sd)
    assert torch.equal(e.encode(a1).cpu(), base)
    e.close()

    sd = OrderedDict(synth_sd)
    bad = synth_sd[key_i].clone()
    bad[3, 5] = 225
    sd[key_i] = bad
    e = _engine()
    from mellow_amd.engine import EngineError
    with pytest.raises(EngineError, match="relative_position_index out of range"):
        e.load_state_dict(sd)
    e.close()

    key_m = spec.ENC + "layers.0.blocks.1.attn_mask"
    sd = OrderedDict(synth_sd)
    sd[key_m] = torch.zeros_like(synth_sd[key_m])          # no shift mask: wrapped-around windows attend across the seam
    e = _engine()
    e.load_state_dict(sd)
    other = e.encode(a1).cpu()
    assert torch.isfinite(other).all() and float((other - base).abs().max()) > 1e-4
    e.close()
