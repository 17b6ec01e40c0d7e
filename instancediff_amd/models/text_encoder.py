"""Frozen context text encoders.

The reference builds `CLIPTextContextEncoder` / `HFContextTextEncoder` from pretrained CLIP / BiomedCLIP
weights (models/drift_noise_model.py:71-90) that are not in the snapshot and cannot be fetched offline;
both are frozen (`requires_grad_(False)`, :74-76,88-89) and are passed to the nets as a forward argument
(:252,257), i.e. they sit OUTSIDE the accelerated hot path.  The nets accept
any module with the reference call signature  `text_encoder(tokens[K,N1], context[B,N2,C]) -> [B,K,embed]`
(_modified_BiomedCLIP.py:863-883).  `StubTextEncoder` is the deterministic stand-in of synthetic
benchmarks/tests; it is only ever used on explicit request (`CLIP_Type: stub`)."""
import math
import os

import torch
import torch.nn as nn


class StubTextEncoder(nn.Module):
    ignores_token_ids = True  # one seeded embedding per class index: the ScoreMapModule's placeholder ids are acceptable here

    def __init__(self, n_cls=5, embed_dim=512, token_embed_dim=512, seed=1236):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.register_buffer("table", torch.randn(n_cls, embed_dim, generator=g))
        self.register_buffer("ctx_proj", torch.randn(token_embed_dim, embed_dim, generator=g) / math.sqrt(token_embed_dim))

    def forward(self, text, context):
        return self.table[None] + (context.mean(dim=1) @ self.ctx_proj)[:, None]


def build_text_encoder(pretrain_path=None, CLIP_Type="CLIP"):
    """Returns (frozen encoder, token_embed_dim) for `CLIP_Type` (models/drift_noise_model.py:70-90).

    Nothing here substitutes weights silently:
      * CLIP_Type == "stub" (the synthetic configuration this repo ships: no pretrained file exists offline) -> the seeded
        `StubTextEncoder`, explicitly requested;
      * any other CLIP_Type needs its pretrained file: a configured path that does not exist raises FileNotFoundError (the
        reference would crash in torch.jit.load, _modified_BiomedCLIP.py:831), and a path that exists raises
        NotImplementedError until the CLIP loader lands -- pass an encoder instance via CLIPDriftModel(text_encoder=...)."""
    if str(CLIP_Type).lower() == "stub":
        enc = StubTextEncoder()
        for p in enc.parameters():
            p.requires_grad_(False)
        enc.eval()
        return enc, 512
    if not pretrain_path:
        raise ValueError(f"CLIP_Type={CLIP_Type!r} needs text_encoder_pretrain_path (use CLIP_Type: stub for the seeded stand-in "
                         "of synthetic runs)")
    if not os.path.exists(str(pretrain_path)):
        raise FileNotFoundError(f"text_encoder_pretrain_path={pretrain_path!r} does not exist: refusing to run CLIP_Type={CLIP_Type!r} "
                                "with random text embeddings (set CLIP_Type: stub to ask for the seeded stand-in explicitly)")
    raise NotImplementedError(
        "loading pretrained CLIP/BiomedCLIP text encoders is not implemented yet (SURVEY.md 8f.1); pass an encoder instance via "
        "CLIPDriftModel(text_encoder=...)")
