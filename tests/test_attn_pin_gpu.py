"""Pins the product's ScoreMapModule decoder path (the kernels behind `ScoreMapModule.context_decode`: fused memory projection
-- compact (C+1)-row pre-image at C=64/128, full 256-row memory at C=256 --, token linears with fused LayerNorm, packed token
self-attention, split-key cross-attention with folded K/V projections, GELU MLP) to outputs of the REAL reference
`ContextDecoder` (models/_modified_BiomedCLIP.py:1194-1244 over :448-478, :520-549), tests/golden/attn_golden.npz, and the
general-M conv-block cross-attention / token attention kernels to the reference `Attention` (:448-478)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from instancediff_amd import ops  # noqa: E402
from instancediff_amd.models.modules.MSM_degEmb_Unet import ContextDecoder_Hierachical, ScoreMapModule  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from attn_fixture_util import ATTN_CASES, DEC_CASES, HIER_CASES, seeded_state  # noqa: E402

DEV = "cuda"


@pytest.fixture(scope="module")
def golden_attn():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "attn_golden.npz"))


def rel_err(got, want):
    want = torch.as_tensor(want).double()
    return float((got.detach().cpu().double() - want).abs().max() / want.abs().max())


@pytest.mark.parametrize("tag", list(DEC_CASES))
def test_smm_decoder_path_matches_reference_context_decoder(golden_attn, tag):
    layers, vdim, hw, seed = DEC_CASES[tag]
    smm = ScoreMapModule(visual_dim=vdim, decoder_layers=layers).to(DEV).eval()
    smm.context_decoder.load_state_dict(seeded_state(smm.context_decoder, seed))
    text = torch.from_numpy(golden_attn[f"{tag}/text"]).to(DEV)
    visual = torch.from_numpy(golden_attn[f"{tag}/visual"])  # [B, N, C] tokens
    B, N, C = visual.shape
    h = int(round(N ** 0.5))
    feat = visual.permute(0, 2, 1).reshape(B, C, h, N // h).contiguous().to(DEV)
    with torch.no_grad():
        out = smm.context_decode(feat, text)
    err = rel_err(out, golden_attn[f"{tag}/out"])
    print(f"{tag}: rel err vs real reference ContextDecoder {err:.2e}")
    assert out.shape == (B, 5, C)
    assert err < 2e-5


@pytest.mark.parametrize("tag", list(HIER_CASES))
def test_smm_decoder_path_matches_reference_hierarchical_decoder(golden_attn, tag):
    """ContextDecoder_Hierachical(if_scale=True, if_flash=False) (:1247-1308 over TransformerDecoderLayer_scaled :552-590): the same
    launches as the plain decoder, the branch gains in the residual linears' epilogue; outdim != visual_dim included"""
    layers, vdim, hw, outdim, seed = HIER_CASES[tag]
    smm = ScoreMapModule(visual_dim=vdim, decoder_layers=layers, decoder_type="ContextDecoder_Hierachical")
    if outdim != vdim:
        smm.context_decoder = ContextDecoder_Hierachical(256, 4, layers, vdim, 512, outdim=outdim)
    smm = smm.to(DEV).eval()
    smm.context_decoder.load_state_dict(seeded_state(smm.context_decoder, seed))
    text = torch.from_numpy(golden_attn[f"{tag}/text"]).to(DEV)
    visual = torch.from_numpy(golden_attn[f"{tag}/visual"])
    B, N, C = visual.shape
    h = int(round(N ** 0.5))
    feat = visual.permute(0, 2, 1).reshape(B, C, h, N // h).contiguous().to(DEV)
    with torch.no_grad():
        out = smm.context_decode(feat, text)
    err = rel_err(out, golden_attn[f"{tag}/out"])
    print(f"{tag}: rel err vs real reference ContextDecoder_Hierachical {err:.2e}")
    assert out.shape == (B, 5, outdim)
    assert err < 2e-5


@pytest.mark.parametrize("tag", list(ATTN_CASES))
def test_token_attention_kernel_matches_reference_attention(golden_attn, tag):
    """reference Attention.forward (:464-478) = q/k/v Linear -> attention core -> proj, on idiff_linear_t + idiff_attn_tokens"""
    dim, heads, N, M, seed = ATTN_CASES[tag]
    sd = {k: v.to(DEV) for k, v in seeded_state(torch.nn.ModuleDict(dict(
        q_proj=torch.nn.Linear(dim, dim, bias=False), k_proj=torch.nn.Linear(dim, dim, bias=False),
        v_proj=torch.nn.Linear(dim, dim, bias=False), proj=torch.nn.Linear(dim, dim))), seed).items()}
    q = torch.from_numpy(golden_attn[f"{tag}/q"]).to(DEV)
    kv = torch.from_numpy(golden_attn[f"{tag}/kv"]).to(DEV)
    B = q.shape[0]
    scale = (dim // heads) ** -0.5
    dh = dim // heads
    qp = ops.linear_t(q.reshape(B * N, dim), sd["q_proj.weight"].t().contiguous())
    if M <= 64:  # token-side attention (ScoreMapModule self-attention kernel)
        kp = ops.linear_t(kv.reshape(B * M, dim), sd["k_proj.weight"].t().contiguous()).reshape(B, M, dim)
        vp = ops.linear_t(kv.reshape(B * M, dim), sd["v_proj.weight"].t().contiguous()).reshape(B, M, dim)
        a = ops.attn_tokens(qp.reshape(B, N, dim), kp, vp, heads, scale).reshape(B * N, dim)
    else:  # few queries, many keys: the ScoreMapModule cross-attention kernel, K/V projections folded onto the queries
        wk, wvT = sd["k_proj.weight"].contiguous(), sd["v_proj.weight"].t().contiguous()
        qf = torch.empty((B * N, heads * dim), device=DEV)
        ops.linear_t_heads(qp, wk, None, qf, heads, dh, dim, x_hs=dh, w_hs=dh * wk.stride(0), b_hs=0, o_hs=dim)
        mem = kv.permute(0, 2, 1).contiguous()  # [B, C, M]: keys on the fast axis, as feature maps are stored
        o = ops.smm_xattn(qf.reshape(B, N, heads, dim), mem, scale).reshape(B * N, heads * dim)
        a = torch.empty((B * N, dim), device=DEV)
        ops.linear_t_heads(o, wvT, None, a, heads, dim, dh, x_hs=dim, w_hs=dh, b_hs=dh, o_hs=dh)
    out = ops.linear_t(a, sd["proj.weight"].t().contiguous(), sd["proj.bias"]).reshape(B, N, dim)
    err = rel_err(out, golden_attn[f"{tag}/out"])
    print(f"{tag}: rel err vs real reference Attention {err:.2e}")
    assert err < 1e-5
