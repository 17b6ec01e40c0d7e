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


# ---- CLIP_Type "BiomedCLIP": HFContextTextEncoder (models/_modified_BiomedCLIP.py:885-1015) -----------------------------------------
from attn_fixture_util import HFTEXT_CASES, HFTEXT_SCALE  # noqa: E402
from instancediff_amd.models.text_encoder import HFContextTextEncoder, build_text_encoder  # noqa: E402


@pytest.fixture(scope="module")
def golden_hftext():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hftext_golden.npz"))


@pytest.mark.parametrize("tag", list(HFTEXT_CASES))
def test_hf_text_encoder_matches_reference(golden_hftext, tag):
    """fixture = the real reference class (constructor, token splice, mask, pooler, projection) around transformers' BertModel.forward
    (tests/golden/make_golden_hftext.py says why that one step comes from the dependency); the product restates all of it in plain
    torch with the reference's parameter paths, so the seeded state dict of the reference class loads unchanged"""
    K, N1, N2, seed = HFTEXT_CASES[tag]
    enc = HFContextTextEncoder(output_tokens=True).eval()
    assert len(enc.state_dict()) == 199  # 5 embedding tensors + 12 x 16 + 2 projection matrices: the reference module's count
    enc.load_state_dict(seeded_state(enc, seed, scale=HFTEXT_SCALE))
    x = torch.from_numpy(golden_hftext[f"{tag}/x"])
    ctx = torch.from_numpy(golden_hftext[f"{tag}/context"])
    assert x.shape == (K, N1) and ctx.shape == (1, N2, 768) and int((x == 0).sum()) > 0
    with torch.no_grad():
        out, hidden = enc(x, ctx)
    want = torch.from_numpy(golden_hftext[f"{tag}/out"])
    want_h = torch.from_numpy(golden_hftext[f"{tag}/cls_hidden"])
    assert out.shape == want.shape == (K, 512)
    eh = float((hidden[:, 0] - want_h).abs().max() / want_h.abs().max())
    eo = float((out - want).abs().max() / want.abs().max())
    assert eh < 2e-5 and eo < 2e-5, (tag, eh, eo)
    # batch of identical context sets: rows b*K + k repeat (the reference's own mask only allows B*K == 5)
    with torch.no_grad():
        rep, _ = enc(x, ctx.expand(2, -1, -1))
    assert rep.shape == (2 * K, 512) and float((rep[:K] - rep[K:]).abs().max()) < 1e-6 and float((rep[:K] - out).abs().max()) < 1e-5


def test_hf_text_encoder_restates_transformers_bert_at_a_small_shape():
    """the transformer inside, against the installed third-party implementation it restates (random small config, padding included)"""
    transformers = pytest.importorskip("transformers")
    torch.manual_seed(3)
    kw = dict(hidden_size=64, num_hidden_layers=3, num_attention_heads=4, intermediate_size=160, vocab_size=120, max_position_embeddings=48)
    bert = transformers.BertModel(transformers.BertConfig(**kw), add_pooling_layer=False).eval()
    enc = HFContextTextEncoder(output_dim=32, output_tokens=True, **kw).eval()
    missing, unexpected = enc.load_state_dict({"transformer." + k: v for k, v in bert.state_dict().items()}, strict=False)
    assert sorted(missing) == ["proj.0.weight", "proj.2.weight"] and not unexpected
    x = torch.randint(1, 119, (5, 9))
    x[1, 5:] = 0
    x[3, 7:] = 0
    ctx = torch.randn(2, 4, 64)
    keep = torch.ones(5, 13).long()
    keep[:, 5:] = (x[:, 1:] != 0).long()
    with torch.no_grad():
        _, h = enc(x, ctx)
        want = bert(inputs_embeds=enc.token_embedding(x, ctx), attention_mask=keep.repeat(2, 1)).last_hidden_state
    assert float((h - want).abs().max()) < 5e-6


def test_build_text_encoder_biomedclip_loads_text_tower(tmp_path):
    """CLIP_Type "BiomedCLIP" (models/drift_noise_model.py:71-77): `text.`-prefixed checkpoint entries load, others are ignored,
    a file without them is refused; token_embed_dim 768"""
    src = HFContextTextEncoder()
    ck = {"text." + k: v.clone() for k, v in src.state_dict().items()}
    ck["visual.trunk.cls_token"] = torch.zeros(1, 1, 768)
    ck["logit_scale"] = torch.tensor(4.6)
    ck["text.transformer.embeddings.position_ids"] = torch.arange(512)[None]  # stored by older transformers versions
    path = tmp_path / "open_clip_pytorch_model.bin"
    torch.save(ck, path)
    enc, dim = build_text_encoder(str(path), "BiomedCLIP")
    assert dim == 768 and isinstance(enc, HFContextTextEncoder) and not enc.training
    assert all(not p.requires_grad for p in enc.parameters())
    for k, v in src.state_dict().items():
        assert torch.equal(enc.state_dict()[k], v), k
    bad = tmp_path / "not_biomedclip.bin"
    torch.save({"visual.x": torch.zeros(2)}, bad)
    with pytest.raises(ValueError, match="text"):
        build_text_encoder(str(bad), "BiomedCLIP")
