"""ORACLE (test infrastructure, not product code).

Plain-PyTorch fp32 CPU restatement of the build's FROZEN SPEC of the score-SDE UNet
`LearnableForwardUNet_MultiScoreMap` and of the `ScoreMapModule`.

** parity unpinned ** -- the reference snapshot does not contain models/modules/MSM_degEmb_Unet.py
(SURVEY.md §0.3), so no reference source, golden vector or checkpoint pins this arithmetic.  What IS
pinned by reference source and restated literally here:
  * the forward contract  net(x_a, x_b, t, names, text_encoder, image_context=None) -> (pred, [sm_0..3])
    (models/drift_noise_model.py:250-268, 234-240)
  * the attention formula  einsum('bnkc,bmkc->bknm')*scale -> softmax(-1) -> einsum('bknm,bmkc->bnkc')
    -> proj  (models/_modified_BiomedCLIP.py:464-478)
  * decoder-layer order / LayerNorm placement (…:543-549), ContextDecoder structure (…:1194-1244) and
    init (trunc_normal std .02, LN 1/0; …:1227-1234)
  * config surface (Configurations/config.yml:106-136).

Attribute names match the product modules one-for-one so `oracle.load_state_dict(product.state_dict())`
works.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

ARTIFACT_TYPES = ['speckle in OCT', 'speckle in ultra sound', 'noise in cryo-EM image', 'noise in low dose CT',
                  'Gaussian noise in MRI']  # config.yml:15


# ---------------------------------------------------------------------------------------------------
# attention exactly as _modified_BiomedCLIP.py:464-478 (inputs already projected to q/k/v)
# ---------------------------------------------------------------------------------------------------
def attention_core(q, k, v, num_heads, scale):
    B, N, C = q.shape
    M = k.shape[1]
    q = q.reshape(B, N, num_heads, C // num_heads)
    k = k.reshape(B, M, num_heads, C // num_heads)
    v = v.reshape(B, M, num_heads, C // num_heads)
    attn = torch.einsum('bnkc,bmkc->bknm', q, k) * scale
    attn = attn.softmax(dim=-1)
    return torch.einsum('bknm,bmkc->bnkc', attn, v).reshape(B, N, C)


def attention_core_flash(q, k, v, num_heads, scale):
    """The attention of Attention_flash (_modified_BiomedCLIP.py:509-513):
        flash_attn_func(clamp(q, -255, 255).half(), clamp(k, ...).half(), clamp(v, ...).half(), softmax_scale=scale).float()
    restated on the CPU.  PARITY UNPINNED for this form: flash_attn (Dao-AILab/flash-attention, unpinned in the reference: a bare import,
    _modified_BiomedCLIP.py:14-20) is not importable here, so the recipe of its forward pass is restated from the FlashAttention-2 paper
    (arXiv:2307.08691, Algorithm 1) and its published kernel: scores and softmax statistics in fp32, the unnormalised probabilities cast
    to the input dtype (fp16) for the second product, fp32 accumulation, the normalised output cast to fp16.  (flash-attn takes the
    probabilities against the RUNNING row maximum of its key blocks; this restatement uses the row's final maximum -- the fp16 roundings
    of the probabilities then differ in their last bit for the blocks seen before the maximum: covered by the tests' tolerance.)"""
    B, N, C = q.shape
    M = k.shape[1]
    h16 = lambda t: t.clamp(min=-255, max=255).to(torch.float16).to(torch.float32)  # noqa: E731
    q = h16(q).reshape(B, N, num_heads, C // num_heads)
    k = h16(k).reshape(B, M, num_heads, C // num_heads)
    v = h16(v).reshape(B, M, num_heads, C // num_heads)
    s = torch.einsum('bnkc,bmkc->bknm', q, k) * scale
    p = torch.exp(s - s.amax(dim=-1, keepdim=True))
    l = p.sum(dim=-1)                                        # [B, heads, N], fp32 probabilities
    o = torch.einsum('bknm,bmkc->bnkc', p.to(torch.float16).to(torch.float32), v) / l.permute(0, 2, 1)[..., None]
    return o.to(torch.float16).to(torch.float32).reshape(B, N, C)


class Attention(nn.Module):  # _modified_BiomedCLIP.py:448-478 (qkv_bias=False, dropouts = 0); flash = True: Attention_flash, :481-517 (same parameters)
    def __init__(self, dim, num_heads=8):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.q_proj = nn.Linear(dim, dim, bias=False)
        self.k_proj = nn.Linear(dim, dim, bias=False)
        self.v_proj = nn.Linear(dim, dim, bias=False)
        self.proj = nn.Linear(dim, dim)
        self.flash = False

    def forward(self, q, k, v):
        core = attention_core_flash if self.flash else attention_core
        x = core(self.q_proj(q), self.k_proj(k), self.v_proj(v), self.num_heads, self.scale)
        return self.proj(x)


class InjectedDropout(nn.Module):
    """nn.Dropout(p) whose masks come from outside: in training mode each call takes the next mask of the class-level queue (a
    0/1 tensor of the input's shape; the tests fill it with the masks the product's Philox stream produced, in call order) and
    returns x * mask / (1 - p).  With an empty queue, in eval() or at p = 0 it is the identity."""
    queue = []

    def __init__(self, p):
        super().__init__()
        self.p = float(p)

    def forward(self, x):
        if not self.training or self.p == 0.0 or not InjectedDropout.queue:
            return x
        mask = InjectedDropout.queue.pop(0)
        assert mask.numel() == x.numel(), (mask.shape, x.shape)
        return x * mask.reshape(x.shape).to(x.dtype) / (1.0 - self.p)


class TransformerDecoderLayer(nn.Module):  # :520-549; dropout sites as the reference: both attentions' proj_drop, the MLP's, the block's
    def __init__(self, d_model, nhead, dropout=0.1):
        super().__init__()
        self.self_attn = Attention(d_model, nhead)
        self.cross_attn = Attention(d_model, nhead)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.sa_drop, self.ca_drop, self.dropout = InjectedDropout(dropout), InjectedDropout(dropout), InjectedDropout(dropout)
        self.mlp = nn.Sequential(nn.Linear(d_model, d_model * 4), nn.GELU(), InjectedDropout(dropout),
                                 nn.Linear(d_model * 4, d_model))

    def forward(self, x, mem):
        q = k = v = self.norm1(x)
        x = x + self.sa_drop(self.self_attn(q, k, v))     # Attention.proj_drop (:476-477)
        q = self.norm2(x)
        x = x + self.ca_drop(self.cross_attn(q, mem, mem))
        x = x + self.dropout(self.mlp(self.norm3(x)))
        return x


class TransformerDecoderLayer_scaled(TransformerDecoderLayer):  # :552-590; if_flash=False: plain Attention, True: Attention_flash (:561-566)
    def __init__(self, d_model, nhead, dropout=0.1, if_flash=False):
        super().__init__(d_model, nhead, dropout)
        self.self_attn.flash = self.cross_attn.flash = bool(if_flash)
        self.gamma_sa = nn.Parameter(torch.ones((1, 1, d_model)) * 1e-1)
        self.gamma_ca = nn.Parameter(torch.ones((1, 1, d_model)) * 1e-1)
        self.gamma_mlp = nn.Parameter(torch.ones((1, 1, d_model)) * 1e-1)

    def forward(self, x, mem):  # :584-590
        q = k = v = self.norm1(x)
        x = x + self.gamma_sa * self.sa_drop(self.self_attn(q, k, v))
        q = self.norm2(x)
        x = x + self.gamma_ca * self.ca_drop(self.cross_attn(q, mem, mem))
        x = x + self.gamma_mlp * self.dropout(self.mlp(self.norm3(x)))
        return x


class ContextDecoder(nn.Module):  # :1194-1244
    layer_cls = TransformerDecoderLayer

    def __init__(self, transformer_width=256, transformer_heads=4, transformer_layers=3, visual_dim=512, text_dim=512, outdim=None):
        super().__init__()
        self.memory_proj = nn.Sequential(nn.LayerNorm(visual_dim), nn.Linear(visual_dim, transformer_width),
                                         nn.LayerNorm(transformer_width))
        self.text_proj = nn.Sequential(nn.LayerNorm(text_dim), nn.Linear(text_dim, transformer_width))
        self.decoder = nn.ModuleList([self.layer_cls(transformer_width, transformer_heads)
                                      for _ in range(transformer_layers)])
        self.out_proj = nn.Sequential(nn.LayerNorm(transformer_width), nn.Linear(transformer_width, visual_dim if outdim is None else outdim))

    def forward(self, text, visual):
        visual = self.memory_proj(visual)
        x = self.text_proj(text)
        for layer in self.decoder:
            x = layer(x, visual)
        return self.out_proj(x)


class ContextDecoder_Hierachical(ContextDecoder):  # :1247-1308, if_scale=True: scaled layers, free output width; if_flash as the layers'
    layer_cls = TransformerDecoderLayer_scaled

    def __init__(self, transformer_width=256, transformer_heads=4, transformer_layers=6, visual_dim=512, text_dim=512, outdim=512, if_flash=False):
        super().__init__(transformer_width, transformer_heads, transformer_layers, visual_dim, text_dim, outdim=outdim)
        for layer in self.decoder:
            layer.self_attn.flash = layer.cross_attn.flash = bool(if_flash)


class ScoreMapModule(nn.Module):
    """text emb [B,K,512] (frozen encoder over class prompts + learnable context) -> MHCA stack over the
    conv feature -> text (+) -> text (x) feature -> score map [B,K,h,w]  (figure LDD_Overall2.png)."""

    def __init__(self, visual_dim=64, CLIP_Type="CLIP", token_embed_dim=512, text_dim=512, n_ctx=8, n_cls=5,
                 prompt_len=10, decoder_layers=3, decoder_width=256, decoder_heads=4, decoder_type="ContextDecoder", if_flash=False):
        super().__init__()
        self.visual_dim = visual_dim
        self.contexts = nn.Parameter(torch.zeros(1, n_ctx, token_embed_dim))
        self.register_buffer("tokens", torch.zeros(n_cls, prompt_len, dtype=torch.long))
        self.text_to_visual = nn.Linear(text_dim, visual_dim)
        if decoder_type == "ContextDecoder":
            self.context_decoder = ContextDecoder(decoder_width, decoder_heads, decoder_layers, visual_dim, text_dim)
        else:
            self.context_decoder = ContextDecoder_Hierachical(decoder_width, decoder_heads, decoder_layers, visual_dim, text_dim, outdim=visual_dim,
                                                              if_flash=if_flash)
        self.gamma = nn.Parameter(torch.ones(visual_dim) * 1e-4)

    def forward(self, feat, text_encoder):
        B, C, H, W = feat.shape
        text = text_encoder(self.tokens, self.contexts.expand(B, -1, -1))  # [B,K,text_dim]
        if text.dim() == 2:  # HFContextTextEncoder returns the flat [B*K, text_dim] (_modified_BiomedCLIP.py:979-991)
            text = text.reshape(B, -1, text.shape[-1])
        vis = feat.reshape(B, C, H * W).permute(0, 2, 1)
        diff = self.context_decoder(text, vis)  # [B,K,C]
        tv = self.text_to_visual(text) + self.gamma * diff
        tv = F.normalize(tv, dim=2, p=2)
        vn = F.normalize(feat, dim=1, p=2)
        return torch.einsum('bchw,bkc->bkhw', vn, tv)


# ---------------------------------------------------------------------------------------------------
# UNet pieces
# ---------------------------------------------------------------------------------------------------
class SinusoidalPosEmb(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, t):
        half = self.dim // 2
        freq = torch.exp(torch.arange(half, dtype=torch.float32, device=t.device) * (-math.log(10000.0) / (half - 1)))
        a = t.to(torch.float32)[:, None] * freq[None, :]
        return torch.cat([a.sin(), a.cos()], dim=-1)


class ResBlock(nn.Module):
    """conv3x3 -> GroupNorm -> FiLM(scale+1, shift from time emb) -> SiLU -> conv3x3 -> GroupNorm -> SiLU, + res."""

    def __init__(self, dim_in, dim_out, time_dim, groups=8):
        super().__init__()
        self.mlp = nn.Linear(time_dim, dim_out * 2)  # applied to SiLU(temb)
        self.conv1 = nn.Conv2d(dim_in, dim_out, 3, padding=1)
        self.norm1 = nn.GroupNorm(groups, dim_out)
        self.conv2 = nn.Conv2d(dim_out, dim_out, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, dim_out)
        self.res_conv = nn.Conv2d(dim_in, dim_out, 1) if dim_in != dim_out else nn.Identity()

    def forward(self, x, temb):
        ss = self.mlp(F.silu(temb))[:, :, None, None]
        scale, shift = ss.chunk(2, dim=1)
        h = self.norm1(self.conv1(x))
        h = F.silu(h * (scale + 1) + shift)
        h = F.silu(self.norm2(self.conv2(h)))
        return h + self.res_conv(x)


class ChanLayerNorm(nn.Module):
    """LayerNorm over the channel dim of an NCHW map (per pixel)."""

    def __init__(self, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))

    def forward(self, x):
        return F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), self.weight, self.bias, 1e-5).permute(0, 3, 1, 2)


class CrossAttention(nn.Module):
    """Q <- feature map, K/V <- image embedding tokens [B,M,ctx_dim] (figure: Conv Block)."""

    def __init__(self, dim, ctx_dim, num_heads=4):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.norm = ChanLayerNorm(dim)
        self.q_proj = nn.Conv2d(dim, dim, 1, bias=False)
        self.k_proj = nn.Linear(ctx_dim, dim, bias=False)
        self.v_proj = nn.Linear(ctx_dim, dim, bias=False)
        self.proj = nn.Conv2d(dim, dim, 1)

    def forward(self, x, ctx):
        B, C, H, W = x.shape
        q = self.q_proj(self.norm(x)).reshape(B, C, H * W).permute(0, 2, 1)
        o = attention_core(q, self.k_proj(ctx), self.v_proj(ctx), self.num_heads, self.scale)
        return self.proj(o.permute(0, 2, 1).reshape(B, C, H, W))


class SelfAttention(nn.Module):
    def __init__(self, dim, num_heads=4):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.norm = ChanLayerNorm(dim)
        self.qkv = nn.Conv2d(dim, dim * 3, 1, bias=False)
        self.proj = nn.Conv2d(dim, dim, 1)

    def forward(self, x):
        B, C, H, W = x.shape
        q, k, v = self.qkv(self.norm(x)).reshape(B, 3, C, H * W).permute(1, 0, 3, 2)
        o = attention_core(q, k, v, self.num_heads, self.scale)
        return self.proj(o.permute(0, 2, 1).reshape(B, C, H, W))


class Downsample(nn.Module):  # pixel-unshuffle(2) + 1x1 conv
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.conv = nn.Conv2d(dim_in * 4, dim_out, 1)

    def forward(self, x):
        return self.conv(F.pixel_unshuffle(x, 2))


class Upsample(nn.Module):  # nearest x2 + 3x3 conv
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.conv = nn.Conv2d(dim_in, dim_out, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2, mode="nearest"))


class SameConv(nn.Module):  # last level: plain 3x3 conv, no resolution change
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.conv = nn.Conv2d(dim_in, dim_out, 3, padding=1)

    def forward(self, x):
        return self.conv(x)


class Level(nn.Module):
    pass


class LearnableForwardUNet_MultiScoreMap(nn.Module):
    def __init__(self, in_nc=2, out_nc=5, nf=64, ch_mult=(1, 2, 4, 4), context_dim=512, text_module="scoremap",
                 score_map_chan=16, if_MultiScoreMap=True, score_map_ch_mult=(1, 1, 2, 4), score_map_ngf=16,
                 use_image_context=False, use_degra_context=False, CLIP_ScoreMapModule=None,
                 artifact_types=ARTIFACT_TYPES, gn_groups=8, attn_heads=4, **_ignored):
        super().__init__()
        self.text_module = text_module
        self.use_image_context = use_image_context
        self.type_map_ind = {n: i for i, n in enumerate(artifact_types)}
        self.depth = len(ch_mult)
        K = len(artifact_types)
        time_dim = nf * 4
        mult = [1] + list(ch_mult)
        self.init_conv = nn.Conv2d(in_nc, nf, 7, padding=3)
        self.time_pos = SinusoidalPosEmb(nf)
        self.time_mlp = nn.Sequential(nn.Linear(nf, time_dim), nn.GELU(), nn.Linear(time_dim, time_dim))
        use_sm = text_module == "scoremap"
        self.CLIP_ScoreMapModule = CLIP_ScoreMapModule if use_sm else None
        # if_MultiScoreMap False (models/drift_noise_model.py:113-114,130-131: ONE default ScoreMapModule() handed to the net; frozen spec,
        # DESIGN.md section 2): the module sits on the full-resolution level, its score map is embedded into score_map_chan channels of
        # that level's skip; the other levels have no score-map channels
        self.n_sm = (self.depth if if_MultiScoreMap else 1) if use_sm else 0
        self.downs = nn.ModuleList()
        self.ups = nn.ModuleList()
        self.sm_embed = nn.ModuleList()
        for i in range(self.depth):
            din, dout = nf * mult[i], nf * mult[i + 1]
            smc = (score_map_ngf * score_map_ch_mult[i] if if_MultiScoreMap else score_map_chan) if i < self.n_sm else 0
            lv = Level()
            lv.res1 = ResBlock(din, din, time_dim, gn_groups)
            lv.res2 = ResBlock(din, din, time_dim, gn_groups)
            if use_image_context:
                lv.ca1 = CrossAttention(din, context_dim, attn_heads)
                lv.ca2 = CrossAttention(din, context_dim, attn_heads)
            lv.down = Downsample(din, dout) if i != self.depth - 1 else SameConv(din, dout)
            self.downs.append(lv)
            if i < self.n_sm:
                self.sm_embed.append(nn.Conv2d(K, smc, 3, padding=1))
            up = Level()
            up.res1 = ResBlock(dout + din + smc, dout, time_dim, gn_groups)
            up.res2 = ResBlock(dout + din, dout, time_dim, gn_groups)
            if use_image_context:
                up.ca1 = CrossAttention(dout, context_dim, attn_heads)
                up.ca2 = CrossAttention(dout, context_dim, attn_heads)
            up.up = Upsample(dout, din) if i != 0 else SameConv(dout, din)
            self.ups.insert(0, up)
        mid = nf * mult[-1]
        self.mid_res1 = ResBlock(mid, mid, time_dim, gn_groups)
        self.mid_attn = SelfAttention(mid, attn_heads)
        if use_image_context:
            self.mid_ca = CrossAttention(mid, context_dim, attn_heads)
        self.mid_res2 = ResBlock(mid, mid, time_dim, gn_groups)
        self.final_res = ResBlock(nf * 2, nf, time_dim, gn_groups)
        self.final_conv = nn.Conv2d(nf, out_nc, 3, padding=1)

    def class_index(self, names, device):
        return torch.tensor([self.type_map_ind[n] for n in names], dtype=torch.long, device=device)

    def forward(self, x_a, x_b, t, names, text_encoder, image_context=None):
        B = x_a.shape[0]
        if not torch.is_tensor(t):
            t = torch.full((B,), float(t), dtype=torch.float32, device=x_a.device)
        t = t.reshape(-1).to(torch.float32)
        if t.numel() == 1 and B > 1:
            t = t.expand(B)
        idx = self.class_index(names, x_a.device)
        ctx = image_context if self.use_image_context else None
        x = self.init_conv(torch.cat([x_a, x_b], dim=1))
        x_ = x
        temb = self.time_mlp(self.time_pos(t))
        h, sms = [], []
        for i, lv in enumerate(self.downs):
            x = lv.res1(x, temb)
            if ctx is not None:
                x = x + lv.ca1(x, ctx)
            h.append(x)
            x = lv.res2(x, temb)
            if ctx is not None:
                x = x + lv.ca2(x, ctx)
            if i < self.n_sm:
                smm = self.CLIP_ScoreMapModule[i] if isinstance(self.CLIP_ScoreMapModule, (nn.ModuleList, list, tuple)) else self.CLIP_ScoreMapModule
                score = smm(x, text_encoder)  # [B,K,h,w]
                sms.append(score[torch.arange(B), idx][:, None])
                h.append(torch.cat([x, self.sm_embed[i](score)], dim=1))
            else:
                h.append(x)
            x = lv.down(x)
        x = self.mid_res1(x, temb)
        x = x + self.mid_attn(x)
        if ctx is not None:
            x = x + self.mid_ca(x, ctx)
        x = self.mid_res2(x, temb)
        for up in self.ups:
            x = up.res1(torch.cat([x, h.pop()], dim=1), temb)
            if ctx is not None:
                x = x + up.ca1(x, ctx)
            x = up.res2(torch.cat([x, h.pop()], dim=1), temb)
            if ctx is not None:
                x = x + up.ca2(x, ctx)
            x = up.up(x)
        x = self.final_res(torch.cat([x, x_], dim=1), temb)
        out = self.final_conv(x)
        pred = out[torch.arange(B), idx][:, None]
        if self.text_module == "scoremap":
            return pred, sms
        return pred


class StubTextEncoder(nn.Module):
    """Frozen stand-in for CLIPTextContextEncoder.forward(text[K,N1], context[B,N2,C]) -> [B,K,512]
    (_modified_BiomedCLIP.py:863-883): a fixed class table plus a linear read-out of the mean context."""

    def __init__(self, n_cls=5, embed_dim=512, token_embed_dim=512, seed=1236):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.register_buffer("table", torch.randn(n_cls, embed_dim, generator=g))
        self.register_buffer("ctx_proj", torch.randn(token_embed_dim, embed_dim, generator=g) / math.sqrt(token_embed_dim))

    def forward(self, text, context):
        return self.table[None] + (context.mean(dim=1) @ self.ctx_proj)[:, None]
