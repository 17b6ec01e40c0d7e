"""Pins the product's frozen context text encoder (`instancediff_amd.models.text_encoder.CLIPTextContextEncoder`) to outputs of
the REAL reference class (models/_modified_BiomedCLIP.py:798-883), tests/golden/text_golden.npz (make_golden_text.py): context
splice after the first token, positional embedding, causal transformer with CLIP's parameter names, end-of-text gather, projection.
Host-side torch module (frozen forward-argument of the nets), so this is a CPU test."""
import os
import sys

import numpy as np
import pytest
import torch

from instancediff_amd.models.text_encoder import CLIPTextContextEncoder

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from attn_fixture_util import TEXT_CASES, seeded_state  # noqa: E402


@pytest.fixture(scope="module")
def golden_text():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "text_golden.npz"))


@pytest.mark.parametrize("tag", list(TEXT_CASES))
def test_text_encoder_matches_reference(golden_text, tag):
    kw, n_ctx, K, B, seed = TEXT_CASES[tag]
    enc = CLIPTextContextEncoder(**kw).eval()
    sd = seeded_state(enc, seed)  # same names as the reference class: the state dict of one loads into the other
    enc.load_state_dict(sd)
    text = torch.from_numpy(golden_text[f"{tag}/text"])
    ctx = torch.from_numpy(golden_text[f"{tag}/context"])
    with torch.no_grad():
        out = enc(text, ctx)
    want = torch.from_numpy(golden_text[f"{tag}/out"])
    assert out.shape == want.shape == (B, K, kw["embed_dim"])
    err = float((out - want).abs().max() / want.abs().max())
    assert err < 5e-6, (tag, err)
    # one context set broadcast over a batch gives identical rows (what ScoreMapModule relies on to run the encoder once)
    with torch.no_grad():
        rep = enc(text, ctx[:1].expand(3, -1, -1))
    assert torch.allclose(rep[0], rep[2], atol=0, rtol=0) or float((rep[0] - rep[2]).abs().max()) < 1e-6


def test_text_encoder_rejects_wrong_lengths():
    enc = CLIPTextContextEncoder(context_length=14, vocab_size=300, transformer_width=64, transformer_heads=4, transformer_layers=1, embed_dim=8)
    with pytest.raises(ValueError, match="context_length"):
        enc(torch.ones(5, 9, dtype=torch.long), torch.zeros(1, 4, 64))
