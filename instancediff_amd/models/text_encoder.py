"""Frozen context text encoders.

The reference builds `CLIPTextContextEncoder` / `HFContextTextEncoder` from pretrained CLIP / BiomedCLIP
weights (models/drift_noise_model.py:71-90) that are not in the snapshot and cannot be fetched offline;
both are frozen (`requires_grad_(False)`, :74-76,88-89) and are passed to the nets as a forward argument
(:252,257), i.e. they sit OUTSIDE the accelerated hot path.  The nets accept
any module with the reference call signature  `text_encoder(tokens[K,N1], context[B,N2,C]) -> [B,K,embed]`
(_modified_BiomedCLIP.py:863-883).  `StubTextEncoder` is the deterministic stand-in used when no
pretrained encoder is supplied (synthetic benchmarks/tests)."""
import math
import os

import torch
import torch.nn as nn


class StubTextEncoder(nn.Module):
    def __init__(self, n_cls=5, embed_dim=512, token_embed_dim=512, seed=1236):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.register_buffer("table", torch.randn(n_cls, embed_dim, generator=g))
        self.register_buffer("ctx_proj", torch.randn(token_embed_dim, embed_dim, generator=g) / math.sqrt(token_embed_dim))

    def forward(self, text, context):
        return self.table[None] + (context.mean(dim=1) @ self.ctx_proj)[:, None]


def build_text_encoder(pretrain_path=None, CLIP_Type="CLIP"):
    """Returns (frozen encoder, token_embed_dim).  A real encoder can be plugged in by the caller through
    CLIPDriftModel(text_encoder=...); without pretrained weights on disk the stub is used."""
    if pretrain_path and os.path.exists(str(pretrain_path)):
        raise NotImplementedError(
            "loading pretrained CLIP/BiomedCLIP text encoders is outside the hot-path scope (SURVEY.md §8f.1); "
            "pass an encoder instance via CLIPDriftModel(text_encoder=...)")
    enc = StubTextEncoder()
    for p in enc.parameters():
        p.requires_grad_(False)
    enc.eval()
    return enc, 512
