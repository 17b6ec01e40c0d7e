"""Shared by tests/golden/make_golden_attn.py (generator, real reference classes) and the tests that replay attn_golden.npz:
the case table and the seeded weights.  Weights are rebuilt from seeds instead of being stored (torch's CPU generator is
platform-stable), so the fixture holds inputs and the reference's outputs only."""
import zlib

import torch

# tag -> (dim, heads, N queries, M keys, weight seed)          Attention, _modified_BiomedCLIP.py:448-478
ATTN_CASES = {"attn_self": (256, 4, 5, 5, 11), "attn_cross": (256, 4, 5, 300, 12), "attn_h8": (64, 8, 7, 33, 13)}
LAYER_SEED = 77  # TransformerDecoderLayer(256, 4), :520-549
# tag -> (decoder layers, visual_dim, visual tokens, weight seed)   ContextDecoder, :1194-1244 (reference default: 6 layers)
DEC_CASES = {"dec_c64": (3, 64, 16 * 16, 167), "dec_c128": (3, 128, 8 * 8, 231), "dec_default6": (6, 256, 6 * 6, 362)}
SCALED_LAYER_SEED = 78  # TransformerDecoderLayer_scaled(256, 4, if_flash=False), :552-590
# tag -> (decoder layers, visual_dim, visual tokens, outdim, weight seed)   ContextDecoder_Hierachical(if_scale=True, if_flash=False), :1247-1308
HIER_CASES = {"hier_c64": (3, 64, 16 * 16, 64, 468), "hier_default6_o512": (6, 128, 8 * 8, 512, 533)}


def seeded_state(module, seed, scale=0.08):
    """state dict with every parameter replaced by seeded values of a useful size (LayerNorm gains around 1, biases non-zero);
    each tensor's stream is keyed by its NAME, so the values do not depend on the order a class registers its sub-modules"""
    sd = {}
    for k, v in module.state_dict().items():
        g = torch.Generator().manual_seed(seed * 1000003 + zlib.crc32(k.encode()))
        if k.endswith("weight") and v.dim() == 1:  # LayerNorm gain
            sd[k] = 1.0 + 0.2 * torch.randn(v.shape, generator=g)
        elif v.dim() == 1:
            sd[k] = 0.1 * torch.randn(v.shape, generator=g)
        else:
            sd[k] = scale * torch.randn(v.shape, generator=g)
    return sd

# tag -> (CLIPTextContextEncoder kwargs, context tokens N2, class prompts K, batch, weight seed)   _modified_BiomedCLIP.py:798-883
TEXT_CASES = {
    "text_small": (dict(context_length=14, vocab_size=300, transformer_width=64, transformer_heads=4, transformer_layers=2, embed_dim=48), 4, 5, 2, 901),
    "text_ref_shape": (dict(context_length=42, vocab_size=49408, transformer_width=512, transformer_heads=8, transformer_layers=12, embed_dim=512),
                       8, 5, 1, 902),  # models/drift_noise_model.py:79-86
}

# tag -> (class prompts K (the reference's mask is built for exactly 5, :966), prompt length N1, context tokens N2, weight seed)
# HFContextTextEncoder, _modified_BiomedCLIP.py:885-1015 (PubMedBERT shape is hard-coded there: one size only)
HFTEXT_CASES = {"hftext_n8": (5, 12, 8, 1201), "hftext_n32": (5, 10, 32, 1202)}
HFTEXT_SCALE = 0.03  # std of the seeded matrices (768-wide layers)
