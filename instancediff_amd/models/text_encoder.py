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


class _Block(nn.Module):
    """One pre-LN transformer block with the parameter names of CLIP's ResidualAttentionBlock (so CLIP checkpoints load):
    x += out_proj(softmax(q k^T / sqrt(dh) + mask) v) over LN1(x);  x += c_proj(QuickGELU(c_fc(LN2(x))))
    (_modified_BiomedCLIP.py:371-408; QuickGELU = x * sigmoid(1.702 x), :323-325)."""

    def __init__(self, width, heads):
        super().__init__()
        self.heads = heads
        self.attn = nn.MultiheadAttention(width, heads)  # parameter container: in_proj_weight / in_proj_bias / out_proj
        self.ln_1 = nn.LayerNorm(width)
        self.mlp = nn.Sequential()
        self.mlp.add_module("c_fc", nn.Linear(width, width * 4))
        self.mlp.add_module("gelu", nn.Identity())
        self.mlp.add_module("c_proj", nn.Linear(width * 4, width))
        self.ln_2 = nn.LayerNorm(width)

    def forward(self, x, mask):  # x [S, L, W] (sequence-major here; the reference runs [L, S, W])
        S, L, W = x.shape
        dh = W // self.heads
        h = self.ln_1(x)
        qkv = h @ self.attn.in_proj_weight.t() + self.attn.in_proj_bias
        q, k, v = [t.reshape(S, L, self.heads, dh).transpose(1, 2) for t in qkv.chunk(3, dim=-1)]  # [S, heads, L, dh]
        att = (q * dh ** -0.5) @ k.transpose(-1, -2) + mask  # nn.MultiheadAttention scales q before the product
        a = (att.softmax(dim=-1) @ v).transpose(1, 2).reshape(S, L, W)
        x = x + a @ self.attn.out_proj.weight.t() + self.attn.out_proj.bias
        m = self.mlp.c_fc(self.ln_2(x))
        return x + self.mlp.c_proj(m * torch.sigmoid(1.702 * m))


class _Transformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.resblocks = nn.Sequential(*[_Block(width, heads) for _ in range(layers)])


class CLIPTextContextEncoder(nn.Module):
    """The frozen context text encoder of the reference (models/_modified_BiomedCLIP.py:798-883; built at
    models/drift_noise_model.py:79-86 with context_length 42, width 512, 8 heads, 12 layers, embed 512):
    `forward(text [K, N1] token ids, context [B, N2, C]) -> [B, K, embed_dim]` -- the learnable context tokens are spliced in after
    the first (start-of-text) token of every class prompt (:869-873), positions added, a causal transformer run, and the feature at
    each prompt's end-of-text position (argmax token id, shifted by N2; :866) projected.  N1 + N2 must equal context_length.
    Parameter names match CLIP's state dict, so `init_weights` loads the text tower of an OpenAI CLIP archive
    (torch.jit.load(...).state_dict(), :829-846).  Frozen, forward-argument of the nets: host-side torch module,
    evaluated once per context set at inference (ScoreMapModule caches it)."""
    ignores_token_ids = False

    def __init__(self, context_length=22, vocab_size=49408, transformer_width=512, transformer_heads=8, transformer_layers=12, embed_dim=1024,
                 out_dim=256, pretrained=None, **kwargs):
        super().__init__()
        self.pretrained, self.context_length, self.embed_dim, self.vocab_size = pretrained, context_length, embed_dim, vocab_size
        self.transformer = _Transformer(transformer_width, transformer_layers, transformer_heads)
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, transformer_width))
        self.ln_final = nn.LayerNorm(transformer_width)
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim))
        nn.init.normal_(self.positional_embedding, std=0.01)
        nn.init.normal_(self.text_projection, std=transformer_width ** -0.5)
        mask = torch.full((context_length, context_length), float("-inf")).triu_(1)  # causal: -inf above the diagonal (:848-854)
        self.register_buffer("attn_mask", mask, persistent=False)

    def init_weights(self, pretrained=None):
        pretrained = pretrained or self.pretrained
        if not isinstance(pretrained, str):
            return
        ck = torch.jit.load(pretrained, map_location="cpu").float().state_dict()
        sd = {}
        for k, v in ck.items():
            if k.startswith("transformer.") or k.startswith("token_embedding") or k.startswith("ln_final") or k in ("positional_embedding", "text_projection"):
                if k == "positional_embedding" and v.size(0) > self.context_length:
                    v = v[:self.context_length]  # 77 -> context_length, as the reference truncates (:839-841)
                sd[k] = v
        missing, unexpected = self.load_state_dict(sd, strict=False)
        if missing or unexpected:
            print(missing, unexpected, "are misaligned params in text encoder")

    def forward(self, text, context):
        x_text = self.token_embedding(text)  # [K, N1, C]
        K, N1, C = x_text.shape
        B, N2, _ = context.shape
        if N1 + N2 != self.context_length:
            raise ValueError(f"prompt length {N1} + context length {N2} must equal context_length {self.context_length}")
        eos = (text.argmax(dim=-1) + N2).reshape(1, K).expand(B, K).reshape(-1)
        x_text = x_text.reshape(1, K, N1, C).expand(B, K, N1, C)
        ctx = context.reshape(B, 1, N2, C).expand(B, K, N2, C)
        x = torch.cat([x_text[:, :, 0:1], ctx, x_text[:, :, 1:]], dim=2).reshape(B * K, N1 + N2, C)
        x = x + self.positional_embedding
        for blk in self.transformer.resblocks:
            x = blk(x, self.attn_mask)
        x = self.ln_final(x)
        x = x[torch.arange(x.shape[0], device=x.device), eos] @ self.text_projection
        return x.reshape(B, K, self.embed_dim)


def _ns(**mods):
    """a bare container whose children carry the given names (for HuggingFace-shaped parameter paths)"""
    m = nn.Module()
    for k, v in mods.items():
        m.add_module(k, v)
    return m


class HFContextTextEncoder(nn.Module):
    """The frozen BiomedCLIP context text encoder of the reference (models/_modified_BiomedCLIP.py:885-1015; selected by
    CLIP_Type "BiomedCLIP", models/drift_noise_model.py:71-77): PubMedBERT = BERT-base (12 post-LN layers x 768, 12 heads, GELU MLP
    3072, vocab 30522, 512 absolute positions, LayerNorm eps 1e-12; the hard-coded config dict :909-915) + CLS last-hidden-state
    pooler + bias-free MLP projection 768 -> 640 -> 512 (:938-944).

    `forward(x [K, N1] token ids, context [B, N2, 768]) -> [B*K, 512]` as the reference returns it (row b*K + k): the context
    tokens are spliced in after the [CLS] embedding of every class prompt (`token_embedding`, :950-958), positions 0..N1+N2-1 and
    token type 0 added, LayerNorm; attention mask = 1 on [CLS] / the context / non-pad prompt tokens (:966-969; the reference
    builds it for exactly 5 prompts and B*K == 5, here K and B are free and the mask repeats over B).  The transformer is restated
    here as plain torch (transformers' BertModel.forward(inputs_embeds=, attention_mask=) in eval mode: the reference's
    modified_BertModel.forward :1081-1191 is a copy of it that ignores its `context` argument) -- no dependency on the
    `transformers` package in the product.  Parameter paths equal the reference module's state dict (`transformer.embeddings.*`,
    `transformer.encoder.layer.<i>.attention.self.query.*`, ..., `proj.0.weight`, `proj.2.weight`), so `init_weights` loads the
    `text.*` entries of a BiomedCLIP checkpoint (open_clip_pytorch_model.bin) exactly as :946-953 does.  Frozen, forward argument
    of the nets: host-side torch module evaluated once per context set."""
    ignores_token_ids = False

    def __init__(self, output_dim=512, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072, vocab_size=30522,
                 max_position_embeddings=512, type_vocab_size=2, pad_token_id=0, layer_norm_eps=1e-12, output_tokens=False):
        super().__init__()
        H = hidden_size
        self.output_dim, self.output_tokens, self.pad_token_id = output_dim, output_tokens, pad_token_id
        self.heads, self.vocab_size, self.context_length = num_attention_heads, vocab_size, max_position_embeddings

        def ln():
            return nn.LayerNorm(H, eps=layer_norm_eps)

        def layer():
            return _ns(attention=_ns(self=_ns(query=nn.Linear(H, H), key=nn.Linear(H, H), value=nn.Linear(H, H)),
                                     output=_ns(dense=nn.Linear(H, H), LayerNorm=ln())),
                       intermediate=_ns(dense=nn.Linear(H, intermediate_size)),
                       output=_ns(dense=nn.Linear(intermediate_size, H), LayerNorm=ln()))
        self.transformer = _ns(
            embeddings=_ns(word_embeddings=nn.Embedding(vocab_size, H, padding_idx=pad_token_id), position_embeddings=nn.Embedding(max_position_embeddings, H),
                           token_type_embeddings=nn.Embedding(type_vocab_size, H), LayerNorm=ln()),
            encoder=_ns(layer=nn.ModuleList([layer() for _ in range(num_hidden_layers)])))
        hid = (H + output_dim) // 2
        self.proj = nn.Sequential(nn.Linear(H, hid, bias=False), nn.GELU(), nn.Linear(hid, output_dim, bias=False))
        for m in self.transformer.modules():  # BertPreTrainedModel._init_weights (initializer_range 0.02)
            if isinstance(m, (nn.Linear, nn.Embedding)):
                nn.init.normal_(m.weight, std=0.02)
                if isinstance(m, nn.Linear):
                    nn.init.zeros_(m.bias)
        with torch.no_grad():
            self.transformer.embeddings.word_embeddings.weight[pad_token_id].zero_()

    def init_weights(self, pretrain_path):
        """load the `text.`-prefixed entries of a BiomedCLIP checkpoint (:946-953; strict=False and the report, as the reference).
        HF buffers some transformers versions store (`embeddings.position_ids`, `embeddings.token_type_ids`) are constants here."""
        state_dict = torch.load(pretrain_path, map_location="cpu", weights_only=True)
        sd = {k[5:]: v for k, v in state_dict.items() if k.startswith("text.")}
        for k in ("transformer.embeddings.position_ids", "transformer.embeddings.token_type_ids"):
            sd.pop(k, None)
        if not sd:
            raise ValueError(f"{pretrain_path!r} holds no `text.*` entries: not a BiomedCLIP checkpoint")
        missing, unexpected = self.load_state_dict(sd, strict=False)
        if len(missing) == len(self.state_dict()):
            raise ValueError(f"none of the text tower's parameters were found in {pretrain_path!r}")
        print(f"{missing}, {unexpected}are misaligned params in text encoder")

    def token_embedding(self, input_ids, context):  # :950-958
        emb = self.transformer.embeddings.word_embeddings(input_ids)
        K, N1, C = emb.shape
        B, N2, _ = context.shape
        emb = emb.reshape(1, K, N1, C).expand(B, K, N1, C)
        ctx = context.reshape(B, 1, N2, C).expand(B, K, N2, C)
        return torch.cat([emb[:, :, 0:1], ctx, emb[:, :, 1:]], dim=2).reshape(B * K, N1 + N2, C)

    def forward(self, x, context):
        K, N1 = x.shape
        B, N2, C = context.shape
        L = N1 + N2
        if L > self.context_length:
            raise ValueError(f"prompt length {N1} + context length {N2} exceeds the {self.context_length} positions")
        keep = torch.ones((K, L), dtype=torch.bool, device=x.device)  # :966-969
        tok = x != self.pad_token_id
        keep[:, 0:1] = tok[:, 0:1]
        keep[:, N2 + 1:] = tok[:, 1:]
        keep = keep.reshape(1, K, L).expand(B, K, L).reshape(B * K, 1, 1, L)
        emb = self.transformer.embeddings
        h = self.token_embedding(x, context).to(emb.LayerNorm.weight.dtype)
        bias = torch.zeros((B * K, 1, 1, L), dtype=h.dtype, device=h.device).masked_fill(~keep, torch.finfo(h.dtype).min)
        h = emb.LayerNorm(h + emb.token_type_embeddings.weight[0] + emb.position_embeddings.weight[:L])
        S, nh = B * K, self.heads
        dh = h.shape[-1] // nh
        for l in self.transformer.encoder.layer:
            sa = getattr(l.attention, "self")
            q, k, v = [f(h).reshape(S, L, nh, dh).transpose(1, 2) for f in (sa.query, sa.key, sa.value)]  # [S, heads, L, dh]
            att = (q @ k.transpose(-1, -2)) * dh ** -0.5 + bias
            a = (att.softmax(dim=-1) @ v).transpose(1, 2).reshape(S, L, nh * dh)
            h = l.attention.output.LayerNorm(l.attention.output.dense(a) + h)
            m = torch.nn.functional.gelu(l.intermediate.dense(h))
            h = l.output.LayerNorm(l.output.dense(m) + h)
        projected = self.proj(h[:, 0])  # ClsLastHiddenStatePooler (BiomedCLIP/hf_model.py:83-93): position 0
        if self.output_tokens:
            return projected, h
        return projected


def build_text_encoder(pretrain_path=None, CLIP_Type="CLIP"):
    """Returns (frozen encoder, token_embed_dim) for `CLIP_Type` (models/drift_noise_model.py:70-90).

    Nothing here substitutes weights silently:
      * CLIP_Type == "stub" (the synthetic configuration this repo ships: no pretrained file exists offline) -> the seeded
        `StubTextEncoder`, explicitly requested;
      * CLIP_Type "CLIP" builds `CLIPTextContextEncoder` and loads the text tower of the configured OpenAI CLIP archive;
        a configured path that does not exist raises FileNotFoundError (the reference would crash in torch.jit.load,
        _modified_BiomedCLIP.py:831);
      * CLIP_Type "BiomedCLIP" builds `HFContextTextEncoder` (PubMedBERT-shaped, token_embed_dim 768) and loads the `text.*`
        entries of the configured BiomedCLIP checkpoint (models/drift_noise_model.py:71-77)."""
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
    if str(CLIP_Type) == "BiomedCLIP":
        enc = HFContextTextEncoder()  # max seq len = 512
        enc.init_weights(pretrain_path=str(pretrain_path))
        for p in enc.parameters():
            p.requires_grad_(False)
        enc.eval()
        return enc, 768
    enc = CLIPTextContextEncoder(context_length=42, embed_dim=512, transformer_width=512, transformer_heads=8, transformer_layers=12,
                                 pretrained=str(pretrain_path))  # models/drift_noise_model.py:79-86
    enc.init_weights()
    for p in enc.parameters():
        p.requires_grad_(False)
    enc.eval()
    return enc, 512
