"""Pins the attention / decoder restatements of oracle/unet_ref.py (`Attention`, `TransformerDecoderLayer`, `ContextDecoder`) to
outputs of the REAL reference classes (models/_modified_BiomedCLIP.py:448-478, 520-549, 1194-1244), captured by
tests/golden/make_golden_attn.py.  Same torch-CPU ops in the same order -> compared at float32 round-off (the reference reshapes
before the einsums exactly as the oracle does; BLAS blocking may differ between hosts, hence a tolerance instead of bit equality)."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import unet_ref

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from attn_fixture_util import ATTN_CASES, DEC_CASES, HIER_CASES, LAYER_SEED, SCALED_LAYER_SEED, seeded_state  # noqa: E402

TOL = 2e-6  # relative to the output's max magnitude


@pytest.fixture(scope="module")
def golden_attn():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "attn_golden.npz"))


def rel_err(got, want):
    want = torch.as_tensor(want).double()
    return float((got.double() - want).abs().max() / want.abs().max())


@pytest.mark.parametrize("tag", list(ATTN_CASES))
def test_attention_matches_reference(golden_attn, tag):
    dim, heads, N, M, seed = ATTN_CASES[tag]
    m = unet_ref.Attention(dim, heads).eval()
    m.load_state_dict(seeded_state(m, seed))
    q, kv = torch.from_numpy(golden_attn[f"{tag}/q"]), torch.from_numpy(golden_attn[f"{tag}/kv"])
    assert q.shape == (2, N, dim) and kv.shape == (2, M, dim)
    with torch.no_grad():
        out = m(q, kv, kv)
    assert rel_err(out, golden_attn[f"{tag}/out"]) < TOL


def test_decoder_layer_matches_reference(golden_attn):
    m = unet_ref.TransformerDecoderLayer(256, 4).eval()
    m.load_state_dict(seeded_state(m, LAYER_SEED))
    with torch.no_grad():
        out = m(torch.from_numpy(golden_attn["layer/x"]), torch.from_numpy(golden_attn["layer/mem"]))
    assert rel_err(out, golden_attn["layer/out"]) < TOL


@pytest.mark.parametrize("tag", list(DEC_CASES))
def test_context_decoder_matches_reference(golden_attn, tag):
    layers, vdim, hw, seed = DEC_CASES[tag]
    m = unet_ref.ContextDecoder(256, 4, layers, vdim, 512).eval()
    m.load_state_dict(seeded_state(m, seed))
    with torch.no_grad():
        out = m(torch.from_numpy(golden_attn[f"{tag}/text"]), torch.from_numpy(golden_attn[f"{tag}/visual"]))
    assert out.shape == (2, 5, vdim)
    assert rel_err(out, golden_attn[f"{tag}/out"]) < TOL


def test_scaled_decoder_layer_matches_reference(golden_attn):
    """TransformerDecoderLayer_scaled(if_flash=False), _modified_BiomedCLIP.py:552-590; gains seeded away from their 0.1 init"""
    m = unet_ref.TransformerDecoderLayer_scaled(256, 4).eval()
    m.load_state_dict(seeded_state(m, SCALED_LAYER_SEED))
    assert tuple(m.gamma_sa.shape) == (1, 1, 256)
    with torch.no_grad():
        out = m(torch.from_numpy(golden_attn["slayer/x"]), torch.from_numpy(golden_attn["slayer/mem"]))
    assert rel_err(out, golden_attn["slayer/out"]) < TOL


@pytest.mark.parametrize("tag", list(HIER_CASES))
def test_hierarchical_context_decoder_matches_reference(golden_attn, tag):
    """ContextDecoder_Hierachical(if_scale=True, if_flash=False), :1247-1308"""
    layers, vdim, hw, outdim, seed = HIER_CASES[tag]
    m = unet_ref.ContextDecoder_Hierachical(256, 4, layers, vdim, 512, outdim=outdim).eval()
    m.load_state_dict(seeded_state(m, seed))
    with torch.no_grad():
        out = m(torch.from_numpy(golden_attn[f"{tag}/text"]), torch.from_numpy(golden_attn[f"{tag}/visual"]))
    assert out.shape == (2, 5, outdim)
    assert rel_err(out, golden_attn[f"{tag}/out"]) < TOL
