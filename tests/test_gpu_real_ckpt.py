"""GPU (-m gpu): tests/validate_real_ckpt.py -- the one-command validation for the real `v0.ckpt`.
With MELLOW_CKPT_DIR set (a networked box) it runs on the real checkpoint and tokenizer; otherwise its `--synthetic` self-test
proves that the command itself works: oracle vs engine tokens in both fp32 modes, top-2 gap histogram, fp8 agreement,
the SmolLM2 config.json comparison on a stand-in file."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_validation_command_runs(tmp_path):
    real = os.environ.get("MELLOW_CKPT_DIR") and os.path.exists(os.path.join(os.environ["MELLOW_CKPT_DIR"], "v0.ckpt"))
    out = tmp_path / "report.json"
    cmd = [sys.executable, os.path.join(ROOT, "tools", "validate_real_ckpt.py"), "--out", str(out)]
    if real:
        cmd += ["--steps", "32", "--pairs", "2"] + (["--tokenizer", os.environ["MELLOW_TOKENIZER_DIR"]] if os.environ.get("MELLOW_TOKENIZER_DIR") else [])
    else:
        # a stand-in config.json with SmolLM2-135M's published values: the comparison code path runs, and must come out equal
        cfg = {"vocab_size": 49152, "hidden_size": 576, "intermediate_size": 1536, "num_hidden_layers": 30, "num_attention_heads": 9,
               "num_key_value_heads": 3, "rms_norm_eps": 1e-05, "rope_theta": 100000, "max_position_embeddings": 8192,
               "tie_word_embeddings": True, "bos_token_id": 0, "eos_token_id": 0, "hidden_act": "silu", "attention_bias": False}
        (tmp_path / "config.json").write_text(json.dumps(cfg))
        # the reference's two fixture clips (decoded PCM committed in tests/golden/example.npz) as the --wav example: 44.1 kHz files
        # through the wrapper's own ingest (resample, tile / crop)
        import wave
        import numpy as np
        g = np.load(os.path.join(ROOT, "tests", "golden", "example.npz"))
        for name, key in (("1.wav", "pcm1"), ("2.wav", "pcm2")):
            with wave.open(str(tmp_path / name), "wb") as w:
                w.setnchannels(1); w.setsampwidth(2); w.setframerate(int(g["sr1"])); w.writeframes(g[key].astype("<i2").tobytes())
        cmd += ["--synthetic", "--steps", "6", "--pairs", "2", "--tokenizer", str(tmp_path),
                "--wav", str(tmp_path / "1.wav"), str(tmp_path / "2.wav"), str(g["prompt"])]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    rep = json.loads(out.read_text())
    assert rep["ok"] and rep["lm_config"]["ok"] is True and rep["checkpoint"]["parameters"] == 167020951
    assert rep["tokens"]["f32"]["equal"] and rep["tokens"]["f32x3"]["equal"]
    assert rep["tokens"]["f32"]["teacher_forced_logits_max_abs_diff"] <= 3e-3
    n_examples = len(rep["oracle"]["examples"])
    assert n_examples == (2 if real else 3)
    assert 0.0 <= rep["fp8"]["position_wise_agreement"] <= 1.0 and sum(rep["oracle"]["top2_gap"]["histogram"].values()) == n_examples * rep["steps"]
