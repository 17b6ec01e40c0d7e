"""`LearnableForwardUNet_MultiScoreMap` and `ScoreMapModule` -- the score-SDE UNet of InstanceDiff on
hand-written gfx950 HIP kernels.

The reference imports these names from models/modules/MSM_degEmb_Unet.py (models/drift_noise_model.py:18,
Configurations/config.yml:107-108) but the file is absent from the reference snapshot; the
architecture implemented here is the build's frozen spec, constrained by the forward
contract (drift_noise_model.py:250-268), config.yml:106-136, figures/LDD_Overall2.png and the attention /
decoder blocks of models/_modified_BiomedCLIP.py:448-590,1194-1244.

nn.Modules here are PARAMETER CONTAINERS (state_dict / optimizer / DDP compatibility, same keys as the
oracle in oracle/unet_ref.py); every arithmetic operation of forward() runs in the kernels behind
include/idiff.h via instancediff_amd.ops.  There is no ATen fallback: on a machine without the built
library or without a GPU, forward() raises.
"""
import math
import os
import weakref

import torch
import torch.nn as nn

from ... import ops

ARTIFACT_TYPES = ['speckle in OCT', 'speckle in ultra sound', 'noise in cryo-EM image', 'noise in low dose CT',
                  'Gaussian noise in MRI']  # Configurations/config.yml:15


def _trunc_normal_(t, std=0.02):
    return nn.init.trunc_normal_(t, std=std, a=-2 * std, b=2 * std)


# ---------------------------------------------------------------------------------------------------
# prepared-weight cache: packed conv weights, concatenated / transposed linear weights.  Keyed on the
# parameters' (data_ptr, _version) so optimizer steps and load_state_dict invalidate it.
# ---------------------------------------------------------------------------------------------------
def _weight_epoch():
    from ... import train_ops
    return train_ops.WEIGHT_EPOCH[0]


class _Prepared:
    """per-owner-module cache (weakly keyed, so it dies with the module and ids can never alias)."""

    def __init__(self):
        self.store = weakref.WeakKeyDictionary()

    def get(self, key, params, build):
        owner, name = key[1], key[0]
        # WEIGHT_EPOCH: the fused Adam kernel rewrites parameters through raw pointers (no _version bump)
        sig = (_weight_epoch(),) + tuple((p.data_ptr(), p._version) for p in params)
        slot = self.store.setdefault(owner, {})
        hit = slot.get(name)
        if hit is not None and hit[0] == sig:
            return hit[1]
        with torch.no_grad():
            val = build()
        slot[name] = (sig, val)
        return val

    def peek(self, key, params):
        """the cached value if it is current, else None (nothing is built)"""
        owner, name = key[1], key[0]
        sig = (_weight_epoch(),) + tuple((p.data_ptr(), p._version) for p in params)
        hit = self.store.get(owner, {}).get(name)
        return hit[1] if hit is not None and hit[0] == sig else None

    def clear(self):
        self.store = weakref.WeakKeyDictionary()


_PREP = _Prepared()

# IDIFF_SMM_SIDE=1 (experiment, off by default): each net's ScoreMapModule phase -- which only feeds the skip connections
# the decoder reads later -- on a stream of its own beside the net's mid blocks.  Capture rule found in r05
# (scripts/proto/capture_fork_probe.py, profiles/r05/x_capture_fork_probe.txt): under HIP-graph capture on ROCm 7.2 a forked stream may
# WAIT on another forked stream, but must itself be joined by the capture's ORIGIN stream -- a forked stream that waits on a stream
# forked later than itself (the "join the side stream back into the net's stream" shape) crashes hipStreamEndCapture.  So the side
# stream is never joined back: it waits for the net's mid blocks and the DECODER continues on it; driftSDE.predict() joins both the
# net's stream and its side stream into the origin.
SMM_SIDE = bool(int(os.environ.get("IDIFF_SMM_SIDE", "0")))
# IDIFF_GROUPED_SMM=0 (A/B runs): the per-level launches of r04 for the memory projection, the cross-attention (+ merge) and the score map
GROUPED_SMM = bool(int(os.environ.get("IDIFF_GROUPED_SMM", "1")))

def packed(conv):
    return _PREP.get(("pk", conv), (conv.weight,), lambda: ops.pack_conv_weight(conv.weight.detach().contiguous()))


def wT(lin):
    """cached transposed weight [K, N] of an nn.Linear (the low-latency token-side linear reads it unit-stride)."""
    return _PREP.get(("wT", lin), (lin.weight,), lambda: lin.weight.detach().t().contiguous())


# ---------------------------------------------------------------------------------------------------
# parameter containers (attribute names == oracle/unet_ref.py)
# ---------------------------------------------------------------------------------------------------
class Attention(nn.Module):  # parameters of _modified_BiomedCLIP.py:448-478 (qkv_bias=False)
    def __init__(self, dim, num_heads=8):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.q_proj = nn.Linear(dim, dim, bias=False)
        self.k_proj = nn.Linear(dim, dim, bias=False)
        self.v_proj = nn.Linear(dim, dim, bias=False)
        self.proj = nn.Linear(dim, dim)


class TransformerDecoderLayer(nn.Module):  # :520-549
    def __init__(self, d_model, nhead):
        super().__init__()
        self.self_attn = Attention(d_model, nhead)
        self.cross_attn = Attention(d_model, nhead)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.mlp = nn.Sequential(nn.Linear(d_model, d_model * 4), nn.GELU(), nn.Identity(), nn.Linear(d_model * 4, d_model))


class TransformerDecoderLayer_scaled(TransformerDecoderLayer):  # :552-590
    """TransformerDecoderLayer with a learnable per-channel gain on each residual branch (x + gamma_sa * self_attn, x + gamma_ca *
    cross_attn, x + gamma_mlp * mlp), initialised to 0.1; parameter names and shapes ([1,1,d]) as the reference's.  The gains ride in
    the residual linears' epilogue (`gscale` of idiff_linear_t_*): no extra launch.  The reference's `if_flash=True` form swaps in
    Attention_flash (:481-517: +-255 clamp, fp16 flash_attn) -- same parameters; here the attention arithmetic is chosen by
    ops.ATTN_DTYPE, and fp32 is what `if_flash=False` computes."""

    def __init__(self, d_model, nhead):
        super().__init__(d_model, nhead)
        self.gamma_sa = nn.Parameter(torch.ones((1, 1, d_model)) * 1e-1)
        self.gamma_ca = nn.Parameter(torch.ones((1, 1, d_model)) * 1e-1)
        self.gamma_mlp = nn.Parameter(torch.ones((1, 1, d_model)) * 1e-1)


def branch_gains(layer):
    """(gamma_sa, gamma_ca, gamma_mlp) as [d] views for a scaled layer, (None, None, None) for a plain one"""
    if hasattr(layer, "gamma_sa"):
        return layer.gamma_sa.reshape(-1), layer.gamma_ca.reshape(-1), layer.gamma_mlp.reshape(-1)
    return None, None, None


class ContextDecoder(nn.Module):  # :1194-1244
    layer_cls = TransformerDecoderLayer
    if_flash = False  # the plain TransformerDecoderLayer has no half-precision form (:520-549)

    def __init__(self, transformer_width=256, transformer_heads=4, transformer_layers=3, visual_dim=512, text_dim=512, dropout=0.1, outdim=None):
        super().__init__()
        self.width, self.heads = transformer_width, transformer_heads
        self.dropout = float(dropout)  # :1200 (training mode only: unet_autograd._smm; the sampling path never drops)
        self.memory_proj = nn.Sequential(nn.LayerNorm(visual_dim), nn.Linear(visual_dim, transformer_width),
                                         nn.LayerNorm(transformer_width))
        self.text_proj = nn.Sequential(nn.LayerNorm(text_dim), nn.Linear(text_dim, transformer_width))
        self.decoder = nn.ModuleList([self.layer_cls(transformer_width, transformer_heads)
                                      for _ in range(transformer_layers)])
        self.out_proj = nn.Sequential(nn.LayerNorm(transformer_width), nn.Linear(transformer_width, visual_dim if outdim is None else outdim))
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):  # :1227-1234
        if isinstance(m, nn.Linear):
            _trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)


class ContextDecoder_Hierachical(ContextDecoder):  # :1247-1308 (if_scale=True, the only form the reference's constructor accepts:
    # its if_scale=False branch passes if_flash= to TransformerDecoderLayer, which has no such argument)
    """ContextDecoder built from TransformerDecoderLayer_scaled, with a free output width `outdim` (reference default 512; a
    ScoreMapModule needs outdim == visual_dim, which is what it passes).  State-dict keys equal the reference class's."""
    layer_cls = TransformerDecoderLayer_scaled

    def __init__(self, transformer_width=256, transformer_heads=4, transformer_layers=6, visual_dim=512, text_dim=512, dropout=0.1, outdim=512,
                 if_flash=False):
        """if_flash: the layers' attentions in the reference's half-precision form (Attention_flash, :481-517 -- the reference class's
        default, True; here False = fp32, what `if_flash=False` computes).  The same parameters either way.  A labelled
        reduced-precision VARIANT, inference only: decoder_tokens_flash below."""
        super().__init__(transformer_width, transformer_heads, transformer_layers, visual_dim, text_dim, dropout, outdim=outdim)
        self.if_flash = bool(if_flash)


DECODER_TYPES = {"ContextDecoder": ContextDecoder, "ContextDecoder_Hierachical": ContextDecoder_Hierachical}


def _class_tokens(n_cls, prompt_len, names=ARTIFACT_TYPES):
    """Placeholder token ids for the class prompts, ONLY meaningful for the stub text encoder (which ignores them): a stable hash
    of the prompt words, EOT (= max id, as CLIP's argmax convention, _modified_BiomedCLIP.py:868) at the end.  A real encoder
    needs real ids: ScoreMapModule(tokenizer=...) / set_class_tokens(), or the `tokens` buffer of a loaded checkpoint."""
    tok = torch.zeros(n_cls, prompt_len, dtype=torch.long)
    for i in range(n_cls):
        words = (names[i] if i < len(names) else f"class {i}").split()
        tok[i, 0] = 49406
        for j, w in enumerate(words[:prompt_len - 2]):
            hsh = 0
            for ch in w:
                hsh = (hsh * 131 + ord(ch)) % 40000
            tok[i, 1 + j] = 1000 + hsh
        tok[i, min(len(words), prompt_len - 2) + 1] = 49407
    return tok


class ScoreMapModule(nn.Module):
    """class prompts (+) learnable context -> frozen text encoder -> text emb; MHCA stack over the conv
    feature (ContextDecoder) added back to the text emb; text (x) feature -> score map [B,K,h,w]."""

    def __init__(self, visual_dim=64, CLIP_Type="CLIP", token_embed_dim=512, text_dim=512, n_ctx=8, n_cls=5, prompt_len=10,
                 decoder_layers=3, decoder_width=256, decoder_heads=4, tokenizer=None, class_names=ARTIFACT_TYPES, dropout=0.1,
                 decoder_type="ContextDecoder", if_flash=False):
        """decoder_type: "ContextDecoder" (the frozen spec) or "ContextDecoder_Hierachical" (scaled layers,
        _modified_BiomedCLIP.py:1247-1308) -- the candidate building blocks SURVEY.md section 8 a6 lists for the missing module.
        if_flash (ContextDecoder_Hierachical only): its attentions in the reference's fp16 form (Attention_flash)."""
        super().__init__()
        self.visual_dim, self.n_cls, self.text_dim = visual_dim, n_cls, text_dim
        self.contexts = nn.Parameter(torch.zeros(1, n_ctx, token_embed_dim))
        _trunc_normal_(self.contexts, std=0.02)
        # class-prompt token ids [K, N1] are a (persistent) buffer: a checkpoint carries them.  `tokenizer` is the hook for real
        # ids -- a callable list[str] -> LongTensor [K, N1] (e.g. clip.tokenize(names, context_length=N1),
        # drift_noise_model.py:78-83); without one the placeholder ids are flagged so a non-stub encoder refuses them.
        self._prompt_len, self._ph_check = prompt_len, None
        self.register_buffer("tokens", _class_tokens(n_cls, prompt_len) if tokenizer is None else
                             torch.as_tensor(tokenizer(list(class_names)[:n_cls])).long())
        self.text_to_visual = nn.Linear(text_dim, visual_dim)
        if decoder_type not in DECODER_TYPES:
            raise ValueError(f"decoder_type {decoder_type!r}: expected one of {sorted(DECODER_TYPES)}")
        if if_flash and decoder_type == "ContextDecoder":
            raise ValueError("if_flash needs decoder_type ContextDecoder_Hierachical: the reference's plain TransformerDecoderLayer has no half-precision form")
        self.context_decoder = DECODER_TYPES[decoder_type](decoder_width, decoder_heads, decoder_layers, visual_dim, text_dim, dropout=dropout,
                                                           **({} if decoder_type == "ContextDecoder" else {"outdim": visual_dim, "if_flash": if_flash}))
        self.gamma = nn.Parameter(torch.ones(visual_dim) * 1e-4)
        self._text_cache = None

    def set_class_tokens(self, tokens):
        """install real class-prompt token ids [K, N1] (from a tokenizer run elsewhere, or a config file)"""
        tokens = torch.as_tensor(tokens).long().to(self.tokens.device)
        assert tokens.dim() == 2 and tokens.shape[0] == self.n_cls, tokens.shape
        self.tokens = tokens
        self._text_cache = None

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        tk = state_dict.get(prefix + "tokens")
        if tk is not None and tk.shape != self.tokens.shape:  # real prompts may be longer than the placeholder ones
            self.tokens = torch.empty_like(tk, device=self.tokens.device)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    @property
    def tokens_are_placeholders(self):
        """True while `tokens` still holds the hash-derived placeholder ids (checked by content, so it survives save/load)"""
        key = (self.tokens.data_ptr(), self.tokens._version)
        if self._ph_check is None or self._ph_check[0] != key:
            ph = _class_tokens(self.n_cls, self._prompt_len)
            self._ph_check = (key, tuple(self.tokens.shape) == tuple(ph.shape) and bool(torch.equal(self.tokens.detach().cpu(), ph)))
        return self._ph_check[1]

    # ---- text branch (frozen encoder: out of the accelerated scope, cached at inference) ---------
    def text_embeddings(self, text_encoder, B):
        if self.tokens_are_placeholders and not getattr(text_encoder, "ignores_token_ids", False):
            raise RuntimeError("ScoreMapModule holds placeholder class-token ids but the text encoder reads token ids: pass "
                               "tokenizer=... / call set_class_tokens(), or load a checkpoint that carries the `tokens` buffer")
        # The reference calls text_encoder(tokens, contexts.expand(B, ...)) (drift_noise_model.py:252): B identical rows.  The
        # encoder runs ONCE on the single context set and its [1,K,D] output is broadcast -- B times less frozen-encoder work per
        # training step, and a sample's text embedding (hence its whole result) cannot depend on the batch it sits in.
        if not torch.is_grad_enabled() or not self.contexts.requires_grad:
            key = (B, self.contexts.data_ptr(), self.contexts._version, _weight_epoch())
            c = self._text_cache
            if c is not None and c[0] == key and c[2]() is text_encoder:
                return c[1]
            with torch.no_grad():
                text = self._encode(text_encoder).expand(B, -1, -1).contiguous()
            self._text_cache = (key, text, weakref.ref(text_encoder))
            return text
        return self._encode(text_encoder).expand(B, -1, -1).contiguous()

    def _encode(self, text_encoder):
        """one context set through the frozen encoder -> [1, K, text_dim]; CLIPTextContextEncoder returns [B, K, D]
        (_modified_BiomedCLIP.py:883), HFContextTextEncoder the flat [B*K, D] (:979-991)"""
        t = text_encoder(self.tokens, self.contexts).float()
        return t.reshape(1, self.n_cls, -1) if t.dim() == 2 else t

    def forward(self, feat, text_encoder, idx=None):
        """feat [B,C,h,w] -> (score [B,K,h,w], sel [B,1,h,w] or None)."""
        B, C, H, W = feat.shape
        text = self.text_embeddings(text_encoder, B)  # [B,K,text_dim]
        # tv = text_to_visual(text) + gamma * out_proj(LN(x))
        tv = self._decoder_tokens(feat, text)
        return ops.scoremap(feat, tv.reshape(B, self.n_cls, C), idx)

    def context_decode(self, feat, text):
        """`ContextDecoder.forward(text, visual)` of the reference (_modified_BiomedCLIP.py:1236-1244) with visual = the tokens of
        `feat` [B,C,h,w]: -> [B,K,C].  The same launches as forward() up to the last linear, which here is the plain out_proj
        (no text_to_visual residual, no gamma); used to pin this path to outputs of the real reference class."""
        B, C = feat.shape[:2]
        out = self._decoder_tokens(feat, text.contiguous(), cache_prefix=False, plain_out=True)  # caller-owned text: not a stable cache key
        return out.reshape(B, text.shape[1], -1)  # outdim wide (== C for ContextDecoder)

    def _decoder_tokens(self, feat, text, cache_prefix=True, plain_out=False):
        """tv [B*K, C] = text_to_visual(text) + gamma * out_proj(LN(x_L)) with x_L the decoder state after the last
        TransformerDecoderLayer (plain_out: out_proj(LN(x_L)) alone): decoder_tokens_grouped for this module alone."""
        return decoder_tokens_grouped([self], [feat], [text], cache_prefix=cache_prefix, plain_out=plain_out)[0]


def decoder_tokens_grouped(smms, feats, texts, cache_prefix=True, plain_out=False):
    """The ContextDecoder token chains of SEVERAL ScoreMapModules (the four UNet levels of a net) advanced in lock step: every
    token-side operation -- the same [B*K, 256]-row linear / attention of each module, with its own weights and, where the level's
    channel count enters, its own shape -- is ONE grouped launch for all modules (idiff_linear_t_grouped_fwd,
    idiff_attn_tokens_grouped_fwd: the descriptors travel in the kernel arguments).  These launches are latency-bound (~10 us each
    whatever the row count), so a net's four chains cost 23 token launches instead of 92.  Per module and layer there remain the
    passes over the feature map: the fused memory projection and the cross-attention (idiff_smm_xattn_fwd).
    Returns, per module, tv [B*K, C] = text_to_visual(text) + gamma * out_proj(LN(x_L))   (plain_out: out_proj(LN(x_L)))."""
    if any(m.context_decoder.if_flash for m in smms):  # the half-precision variant: module by module, unfolded keys / values
        assert all(m.context_decoder.if_flash for m in smms)  # (the training step refuses the variant: unet_autograd.forward_train)
        return [decoder_tokens_flash(m, f, t, plain_out=plain_out) for m, f, t in zip(smms, feats, texts)]
    L = len(smms)
    dec0 = smms[0].context_decoder
    Wd, heads, nlayers = dec0.width, dec0.heads, len(dec0.decoder)
    dh = Wd // heads
    for m in smms:
        d = m.context_decoder
        assert (d.width, d.heads, len(d.decoder)) == (Wd, heads, nlayers), "grouped ScoreMapModules must share the decoder geometry"
    dev = feats[0].device
    B = feats[0].shape[0]
    K = texts[0].shape[1]
    R = B * K

    st = []  # per module: dict(dec, C, Cm, compact, mem, mp)
    by_c = {}  # compact levels of equal channel count (and LayerNorm eps) share ONE memory-projection launch
    for m, feat, text in zip(smms, feats, texts):
        Bm, C, H, W = feat.shape
        assert Bm == B and text.shape[1] == K
        dec = m.context_decoder
        mp = dec.memory_proj
        # Narrow feature maps: LN_256(W.xhat + b) = g2 * ((Wc.xhat + bc) * rstd) + b2 is an affine image of the (C+1)-vector
        # m = [xhat*rstd ; rstd], so the cross-attention streams m (Cm = 72 / 136 rows) instead of the 256-row memory and
        # g2.[Wc|bc] is folded into its query / value projections (b2 drops out of the softmax and returns as a bias).
        Cm = Wd if (C + 1 > 136 or C not in (64, 128)) else (72 if C + 1 <= 72 else 136)  # the compact kernel holds C / 4 channels per thread
        compact = Cm < Wd
        mem = None
        if compact:
            gram, hvec, evar = _PREP.get(("mpvar", mp[1]), (mp[1].weight, mp[1].bias), lambda mp=mp: ops.memory_variance_form(mp[1].weight, mp[1].bias))
            by_c.setdefault((C, mp[0].eps, mp[2].eps), []).append(
                (len(st), dict(feat=feat, ln1_g=mp[0].weight, ln1_b=mp[0].bias, gram=gram, hvec=hvec, evar=evar, Cm=Cm)))
        else:
            wmp = _PREP.get(("mp", mp[1]), (mp[1].weight,),
                            lambda mp=mp, C=C: ops.pack_conv_weight(mp[1].weight.detach().reshape(Wd, C, 1, 1).contiguous()))
            mem = ops.smm_memproj(feat, mp[0].weight, mp[0].bias, wmp, mp[1].bias, mp[2].weight, mp[2].bias, eps=mp[0].eps)
        st.append(dict(m=m, dec=dec, C=C, Cm=Cm, compact=compact, mem=mem, mp=mp, t2d=text.reshape(R, m.text_dim)))
    for (C, e1, e2), items in by_c.items():
        if GROUPED_SMM and len(items) > 1:
            mems = ops.smm_memproj_compact_grouped([it for _, it in items], eps1=e1, eps2=e2)
        else:
            mems = [ops.smm_memproj_compact(it["feat"], it["ln1_g"], it["ln1_b"], it["gram"], it["hvec"], it["evar"], it["Cm"], eps1=e1, eps2=e2)
                    for _, it in items]
        for (i, _), mem in zip(items, mems):
            st[i]["mem"] = mem

    def fold_weights(s, ca):
        mp, Cm = s["mp"], s["Cm"]
        if s["compact"]:
            return _PREP.get(("xfold", ca, Cm), (mp[1].weight, mp[1].bias, mp[2].weight, mp[2].bias, ca.k_proj.weight, ca.v_proj.weight),
                             lambda: _fold_memory_affine(mp[1], mp[2], ca, Cm))
        return ca.k_proj.weight, wT(ca.v_proj), None  # [Wd(dh blocks), Wd], [Wd (c), Wd (n)]

    def self_attn_and_query(sel, xs, li):
        """modules sel (indices into st), states xs -> (xs + self-attention, folded cross-attention queries qf [B*K, heads*Cm]);
        nothing here sees the image.  Five grouped launches."""
        lay = [st[i]["dec"].decoder[li] for i in sel]
        qkv = ops.linear_t_grouped([dict(x=x, wT=_PREP.get(("qkvT", l.self_attn), (l.self_attn.q_proj.weight, l.self_attn.k_proj.weight, l.self_attn.v_proj.weight),
                                                             lambda sa=l.self_attn: torch.cat([sa.q_proj.weight, sa.k_proj.weight, sa.v_proj.weight], 0).detach().t().contiguous()),
                                         ln=(l.norm1.weight, l.norm1.bias, l.norm1.eps)) for x, l in zip(xs, lay)])
        att = ops.attn_tokens_packed_grouped([q.reshape(B, K, 3 * Wd) for q in qkv], heads, lay[0].self_attn.scale)
        xs = ops.linear_t_grouped([dict(x=a.reshape(R, Wd), wT=wT(l.self_attn.proj), bias=l.self_attn.proj.bias, res=x, gscale=branch_gains(l)[0])
                                   for a, l, x in zip(att, lay, xs)])
        qc = ops.linear_t_grouped([dict(x=x, wT=wT(l.cross_attn.q_proj), ln=(l.norm2.weight, l.norm2.bias, l.norm2.eps)) for x, l in zip(xs, lay)])
        # cross attention, k/v projections folded onto the (few) queries, per head:
        #   qf[:, h-block] = qc[:, h-block] @ wkf[h row block]   (Wk's row block IS the transposed-weight form [K=dh][N=Cm])
        qfs, groups = [], []
        for i, q, l in zip(sel, qc, lay):
            Cm = st[i]["Cm"]
            wkf = fold_weights(st[i], l.cross_attn)[0]
            qf = torch.empty((R, heads * Cm), device=dev, dtype=torch.float32)
            qfs.append(qf)
            groups += [dict(x=q[:, h * dh:(h + 1) * dh], wT=wkf[h * dh:(h + 1) * dh], out=qf[:, h * Cm:(h + 1) * Cm]) for h in range(heads)]
        _grouped_chunks(groups)
        return xs, qfs

    def text_prefix(sel):
        """Everything up to the first cross-attention depends on the text embeddings and weights only: text_proj, the first layer's
        self-attention and folded queries, and text_to_visual.  Cached across denoising steps."""
        x0 = ops.linear_t_grouped([dict(x=st[i]["t2d"], wT=wT(st[i]["dec"].text_proj[1]), bias=st[i]["dec"].text_proj[1].bias,
                                        ln=(st[i]["dec"].text_proj[0].weight, st[i]["dec"].text_proj[0].bias, st[i]["dec"].text_proj[0].eps)) for i in sel])
        x1, qf1 = self_attn_and_query(sel, x0, 0)
        t2v = ops.linear_t_grouped([dict(x=st[i]["t2d"], wT=wT(st[i]["m"].text_to_visual), bias=st[i]["m"].text_to_visual.bias) for i in sel])
        return list(zip(x1, qf1, t2v))

    # ---- prefix: from the per-module cache where it is valid, the rest computed together ------------------------------------------
    def prefix_params(s, text):
        dec, m, mp = s["dec"], s["m"], s["mp"]
        l0 = dec.decoder[0]
        return [text, dec.text_proj[0].weight, dec.text_proj[0].bias, dec.text_proj[1].weight, dec.text_proj[1].bias, l0.norm1.weight, l0.norm1.bias,
                l0.self_attn.q_proj.weight, l0.self_attn.k_proj.weight, l0.self_attn.v_proj.weight, l0.self_attn.proj.weight, l0.self_attn.proj.bias,
                l0.norm2.weight, l0.norm2.bias, l0.cross_attn.q_proj.weight, l0.cross_attn.k_proj.weight, mp[1].weight, mp[1].bias, mp[2].weight,
                m.text_to_visual.weight, m.text_to_visual.bias] + ([l0.gamma_sa] if hasattr(l0, "gamma_sa") else [])

    if torch.is_grad_enabled() or not cache_prefix:
        pre = text_prefix(list(range(L)))
    else:
        pre = [None] * L
        todo = []
        for i, (s, text) in enumerate(zip(st, texts)):
            hit = _PREP.peek(("prefix", s["m"], s["Cm"]), prefix_params(s, text))
            if hit is None:
                todo.append(i)
            else:
                pre[i] = hit
        if todo:
            for i, val in zip(todo, text_prefix(todo)):
                pre[i] = _PREP.get(("prefix", st[i]["m"], st[i]["Cm"]), prefix_params(st[i], texts[i]), lambda val=val: val)
    xs = [p[0] for p in pre]
    qfs = [p[1] for p in pre]
    t2vs = [p[2] for p in pre]

    every = list(range(L))
    for li in range(nlayers):
        lay = [s["dec"].decoder[li] for s in st]
        if li > 0:
            xs, qfs = self_attn_and_query(every, xs, li)
        avs, groups = [], []
        # the cross-attentions of all levels in ONE attention launch + ONE merge launch (one scale: the decoders share their geometry)
        assert all(l.cross_attn.scale == lay[0].cross_attn.scale for l in lay)
        if GROUPED_SMM and len(st) > 1:
            os_ = ops.smm_xattn_grouped([qf.reshape(B, K, heads, s["Cm"]) for s, qf in zip(st, qfs)], [s["mem"] for s in st], lay[0].cross_attn.scale)
        else:
            os_ = [ops.smm_xattn(qf.reshape(B, K, heads, s["Cm"]), s["mem"], lay[0].cross_attn.scale) for s, qf in zip(st, qfs)]
        for s, l, o in zip(st, lay, os_):
            Cm, ca = s["Cm"], l.cross_attn
            _, wvf, bvf = fold_weights(s, ca)
            o = o.reshape(R, heads * Cm)
            # per head: av[:, h-block] = o[:, h-block] @ wvf[:, h column block] (+ bvf[h-block])  ([Cm, dh])
            av = torch.empty((R, Wd), device=dev, dtype=torch.float32)
            avs.append(av)
            groups += [dict(x=o[:, h * Cm:(h + 1) * Cm], wT=wvf[:, h * dh:(h + 1) * dh], bias=None if bvf is None else bvf[h * dh:(h + 1) * dh],
                            out=av[:, h * dh:(h + 1) * dh]) for h in range(heads)]
        _grouped_chunks(groups)
        xs = ops.linear_t_grouped([dict(x=av, wT=wT(l.cross_attn.proj), bias=l.cross_attn.proj.bias, res=x, gscale=branch_gains(l)[1])
                                   for av, l, x in zip(avs, lay, xs)])
        hm = ops.linear_t_grouped([dict(x=x, wT=wT(l.mlp[0]), bias=l.mlp[0].bias, act_out=ops.ACT_GELU, ln=(l.norm3.weight, l.norm3.bias, l.norm3.eps))
                                   for x, l in zip(xs, lay)])
        xs = ops.linear_t_grouped([dict(x=h_, wT=wT(l.mlp[3]), bias=l.mlp[3].bias, res=x, gscale=branch_gains(l)[2]) for h_, l, x in zip(hm, lay, xs)])
    if plain_out:
        return ops.linear_t_grouped([dict(x=x, wT=wT(s["dec"].out_proj[1]), bias=s["dec"].out_proj[1].bias,
                                          ln=(s["dec"].out_proj[0].weight, s["dec"].out_proj[0].bias, s["dec"].out_proj[0].eps)) for x, s in zip(xs, st)])
    # residual, per-column gain and LayerNorm fused into the last linear
    return ops.linear_t_grouped([dict(x=x, wT=wT(s["dec"].out_proj[1]), bias=s["dec"].out_proj[1].bias, res=t2v, gscale=s["m"].gamma,
                                      ln=(s["dec"].out_proj[0].weight, s["dec"].out_proj[0].bias, s["dec"].out_proj[0].eps)) for x, s, t2v in zip(xs, st, t2vs)])


def decoder_tokens_flash(m, feat, text, plain_out=False):
    """decoder_tokens_grouped for ONE ScoreMapModule whose ContextDecoder_Hierachical runs its attentions in the reference's
    half-precision form (TransformerDecoderLayer_scaled(if_flash=True) -> Attention_flash, _modified_BiomedCLIP.py:481-517,552-590):
    every layer's q / k / v go through the clamp to +-255 and the fp16 rounding, so the keys and values are materialised --
    k = k_proj(mem), v = v_proj(mem) as 1x1 convs over the full 256-row memory, [B, 256, N] each and per layer -- instead of being
    folded onto the queries.  3 x 2 passes over a [B, 256, N] tensor more than the fp32 path: the variant exists for the reference's
    option surface (BASELINE c5 "fp16 MFMA attention"), not for speed.  Linears, LayerNorms, the MLP and the gains are the fp32 path's."""
    dec = m.context_decoder
    Wd, heads = dec.width, dec.heads
    assert Wd == 256 and heads == 4, "the half-precision cross-attention kernel is built for 4 heads x 64"
    B, C, H, W = feat.shape
    K = text.shape[1]
    R = B * K
    mp = dec.memory_proj
    wmp = _PREP.get(("mp", mp[1]), (mp[1].weight,), lambda: ops.pack_conv_weight(mp[1].weight.detach().reshape(Wd, C, 1, 1).contiguous()))
    mem = ops.smm_memproj(feat, mp[0].weight, mp[0].bias, wmp, mp[1].bias, mp[2].weight, mp[2].bias, eps=mp[0].eps).reshape(B, Wd, H, W)
    t2d = text.reshape(R, m.text_dim)
    x = ops.linear_t(t2d, wT(dec.text_proj[1]), bias=dec.text_proj[1].bias, ln=(dec.text_proj[0].weight, dec.text_proj[0].bias, dec.text_proj[0].eps))
    for l in dec.decoder:
        g_sa, g_ca, g_mlp = branch_gains(l)
        sa, ca = l.self_attn, l.cross_attn
        qkvT = _PREP.get(("qkvT", sa), (sa.q_proj.weight, sa.k_proj.weight, sa.v_proj.weight),
                         lambda sa=sa: torch.cat([sa.q_proj.weight, sa.k_proj.weight, sa.v_proj.weight], 0).detach().t().contiguous())
        qkv = ops.linear_t(x, qkvT, ln=(l.norm1.weight, l.norm1.bias, l.norm1.eps))
        att = ops.attn_tokens_packed_f16(qkv.reshape(B, K, 3 * Wd), heads, sa.scale)
        x = ops.linear_t(att.reshape(R, Wd), wT(sa.proj), bias=sa.proj.bias, res=x, gscale=g_sa)
        qc = ops.linear_t(x, wT(ca.q_proj), ln=(l.norm2.weight, l.norm2.bias, l.norm2.eps))
        wk = _PREP.get(("kconv", ca), (ca.k_proj.weight,), lambda ca=ca: ops.pack_conv_weight(ca.k_proj.weight.detach().reshape(Wd, Wd, 1, 1).contiguous()))
        wv = _PREP.get(("vconv", ca), (ca.v_proj.weight,), lambda ca=ca: ops.pack_conv_weight(ca.v_proj.weight.detach().reshape(Wd, Wd, 1, 1).contiguous()))
        kk = ops.conv2d(mem, wk, None, 1, Wd).reshape(B, Wd, H * W)
        vv = ops.conv2d(mem, wv, None, 1, Wd).reshape(B, Wd, H * W)
        av = ops.smm_xattn_kv_f16(qc.reshape(B, K, Wd), kk, vv, heads, ca.scale)
        x = ops.linear_t(av.reshape(R, Wd), wT(ca.proj), bias=ca.proj.bias, res=x, gscale=g_ca)
        hm = ops.linear_t(x, wT(l.mlp[0]), bias=l.mlp[0].bias, act_out=ops.ACT_GELU, ln=(l.norm3.weight, l.norm3.bias, l.norm3.eps))
        x = ops.linear_t(hm, wT(l.mlp[3]), bias=l.mlp[3].bias, res=x, gscale=g_mlp)
    op = dec.out_proj
    if plain_out:
        return ops.linear_t(x, wT(op[1]), bias=op[1].bias, ln=(op[0].weight, op[0].bias, op[0].eps))
    t2v = ops.linear_t(t2d, wT(m.text_to_visual), bias=m.text_to_visual.bias)
    return ops.linear_t(x, wT(op[1]), bias=op[1].bias, res=t2v, gscale=m.gamma, ln=(op[0].weight, op[0].bias, op[0].eps))


def _scoremaps(feats, tvs, idx):
    """score maps of all levels: ONE grouped launch where every level has 16-byte rows (the streaming form), single launches otherwise"""
    ok = all(f.shape[2] * f.shape[3] % 4 == 0 and f.data_ptr() % 16 == 0 and (f.stride(0) % 4 == 0 or f.shape[0] == 1) for f in feats)
    if GROUPED_SMM and ok and 1 < len(feats) <= ops._lib.SCOREMAP_MAX_GROUPS and len({tv.shape[1] for tv in tvs}) == 1:
        return ops.scoremap_grouped(feats, tvs, idx)
    return [ops.scoremap(f, tv, idx) for f, tv in zip(feats, tvs)]


def _grouped_chunks(groups):
    """a grouped launch takes at most 16 problems (the heads of four modules); longer lists go in as many launches as needed"""
    n = ops._lib.LINEAR_MAX_GROUPS
    for i in range(0, len(groups), n):
        ops.linear_t_grouped(groups[i:i + n])


def _fold_memory_affine(lin, ln2, ca, Cm):
    """Weights of the compact cross-attention (host-side weight preparation, fp64): with P = g2 * [Wc | bc | 0] ([Wd, Cm];
    Wc, bc = the memory Linear's weight / bias centred over its Wd outputs)
        wkf = Wk @ P            [Wd (dh row blocks), Cm]   query fold:  qf_h = q_h @ wkf[h-block]
        wvf = P^T @ Wv^T        [Cm, Wd]                   value fold:  av_h = o_h @ wvf[:, h-block] + bvf[h-block]
        bvf = b2 @ Wv^T         [Wd]                       (softmax weights sum to 1)"""
    W = lin.weight.detach().double()
    b = lin.bias.detach().double()
    Wd, C = W.shape
    P = torch.zeros((Wd, Cm), dtype=torch.float64, device=W.device)
    P[:, :C] = W - W.mean(dim=0, keepdim=True)
    P[:, C] = b - b.mean()
    P *= ln2.weight.detach().double()[:, None]
    WvT = ca.v_proj.weight.detach().double().t()
    wkf = (ca.k_proj.weight.detach().double() @ P).float().contiguous()
    wvf = (P.t() @ WvT).float().contiguous()
    bvf = (ln2.bias.detach().double() @ WvT).float().contiguous()
    return wkf, wvf, bvf


class ResBlock(nn.Module):
    def __init__(self, dim_in, dim_out, time_dim, groups=8):
        super().__init__()
        self.dim_in, self.dim_out, self.groups = dim_in, dim_out, groups
        self.mlp = nn.Linear(time_dim, dim_out * 2)
        self.conv1 = nn.Conv2d(dim_in, dim_out, 3, padding=1)
        self.norm1 = nn.GroupNorm(groups, dim_out)
        self.conv2 = nn.Conv2d(dim_out, dim_out, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, dim_out)
        self.res_conv = nn.Conv2d(dim_in, dim_out, 1) if dim_in != dim_out else nn.Identity()

    def run(self, src0, src1, film, vec=None, out=None):
        """conv3x3 -> GN -> FiLM -> SiLU -> conv3x3 -> GN -> SiLU, + res(src) [+ vec]; src = cat(src0, src1)."""
        B, _, H, W = src0.shape
        Co, G, HW = self.dim_out, self.groups, H * W
        # GroupNorm finalize (statistics -> per-(sample, channel) affine, FiLM folded in) rides on the conv that produces the statistics
        # (one C call; the finalize is a launch enqueued behind the conv)
        h1, ab1 = ops.conv2d(src0, packed(self.conv1), self.conv1.bias, 3, Co, src1=src1,
                             gn=dict(groups=G, gamma=self.norm1.weight, beta=self.norm1.bias, film=film, eps=self.norm1.eps))
        h2, (a2, b2) = ops.conv2d(h1, packed(self.conv2), self.conv2.bias, 3, Co, pro=ab1,
                                  gn=dict(groups=G, gamma=self.norm2.weight, beta=self.norm2.bias, eps=self.norm2.eps))
        if isinstance(self.res_conv, nn.Identity):
            assert src1 is None
            return ops.affine_silu_add(h2, (a2, b2), res=src0, vec=vec, out=out)
        return ops.conv2d(src0, packed(self.res_conv), self.res_conv.bias, 1, Co, src1=src1, aux=(h2, a2, b2), vec=vec, out=out)


class ChanLayerNorm(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class CrossAttention(nn.Module):
    """Q <- feature map, K/V <- image-embedding tokens [B,M,ctx_dim]."""

    def __init__(self, dim, ctx_dim, num_heads=4):
        super().__init__()
        self.dim, self.num_heads = dim, num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.norm = ChanLayerNorm(dim)
        self.q_proj = nn.Conv2d(dim, dim, 1, bias=False)
        self.k_proj = nn.Linear(ctx_dim, dim, bias=False)
        self.v_proj = nn.Linear(ctx_dim, dim, bias=False)
        self.proj = nn.Conv2d(dim, dim, 1)

    def single_token_vec(self, ctx):
        """M == 1: softmax over one key is exactly 1, so the block's output is proj(v_proj(ctx)) at every
        pixel: a per-(sample, channel) vector [B, C] (exact, not an approximation)."""
        B = ctx.shape[0]
        v = ops.linear(ctx.reshape(B, -1), self.v_proj.weight)
        return ops.linear(v, self.proj.weight.reshape(self.dim, self.dim), self.proj.bias)

    def run(self, x, ctx, out=None):
        """general M: returns x + proj(attn(q_proj(norm(x)), k_proj(ctx), v_proj(ctx)))."""
        B, C, H, W = x.shape
        M = ctx.shape[1]
        xn = ops.chan_layernorm(x, self.norm.weight, self.norm.bias)
        q = ops.conv2d(xn, packed(self.q_proj), None, 1, C)
        c2 = ctx.reshape(B * M, -1)
        k = ops.linear(c2, self.k_proj.weight).reshape(B, M, C)
        v = ops.linear(c2, self.v_proj.weight).reshape(B, M, C)
        o = ops.attn_ctx(q, k, v, self.num_heads, self.scale)
        return ops.conv2d(o, packed(self.proj), self.proj.bias, 1, C, res=x, out=out)


class SelfAttention(nn.Module):
    def __init__(self, dim, num_heads=4):
        super().__init__()
        self.dim, self.num_heads = dim, num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.norm = ChanLayerNorm(dim)
        self.qkv = nn.Conv2d(dim, dim * 3, 1, bias=False)
        self.proj = nn.Conv2d(dim, dim, 1)

    def run(self, x, vec=None):
        xn = ops.chan_layernorm(x, self.norm.weight, self.norm.bias)
        qkv = ops.conv2d(xn, packed(self.qkv), None, 1, 3 * self.dim)
        o = ops.attn_self(qkv, self.num_heads, self.scale)
        return ops.conv2d(o, packed(self.proj), self.proj.bias, 1, self.dim, res=x, vec=vec)


class Downsample(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.dim_out = dim_out
        self.conv = nn.Conv2d(dim_in * 4, dim_out, 1)

    def run(self, x):
        return ops.conv2d(x, packed(self.conv), self.conv.bias, 1, self.dim_out, mode=ops.CONV_UNSHUFFLE2)


class Upsample(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.dim_out = dim_out
        self.conv = nn.Conv2d(dim_in, dim_out, 3, padding=1)

    def run(self, x):
        return ops.conv2d(x, packed(self.conv), self.conv.bias, 3, self.dim_out, mode=ops.CONV_UPSAMPLE2)


class SameConv(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.dim_out = dim_out
        self.conv = nn.Conv2d(dim_in, dim_out, 3, padding=1)

    def run(self, x):
        return ops.conv2d(x, packed(self.conv), self.conv.bias, 3, self.dim_out)


class Level(nn.Module):
    pass


class LearnableForwardUNet_MultiScoreMap(nn.Module):
    """forward(x_a, x_b, t, names, text_encoder, image_context=None) -> (pred [B,1,H,W], [sm_0..sm_3])
    (text_module == 'scoremap') or pred -- the contract of models/drift_noise_model.py:250-268."""

    def __init__(self, in_nc=2, out_nc=5, nf=64, ch_mult=(1, 2, 4, 4), context_dim=512, text_module="scoremap", score_map_chan=16,
                 if_MultiScoreMap=True, score_map_ch_mult=(1, 1, 2, 4), score_map_ngf=16, use_image_context=False,
                 use_degra_context=False, CLIP_ScoreMapModule=None, artifact_types=ARTIFACT_TYPES, gn_groups=8, attn_heads=4,
                 **_ignored):
        super().__init__()
        if use_degra_context:
            raise NotImplementedError("use_degra_context=True is outside the hot-path scope (config.yml:133 sets it False)")
        self.text_module = text_module
        self.use_image_context = use_image_context
        self.type_map_ind = {n: i for i, n in enumerate(artifact_types)}
        self.depth = len(ch_mult)
        self.nf, self.out_nc, self.in_nc = nf, out_nc, in_nc
        K = len(artifact_types)
        time_dim = nf * 4
        self.time_dim = time_dim
        mult = [1] + list(ch_mult)
        self.init_conv = nn.Conv2d(in_nc, nf, 7, padding=3)
        self.time_mlp = nn.Sequential(nn.Linear(nf, time_dim), nn.GELU(), nn.Linear(time_dim, time_dim))
        half = nf // 2
        self.register_buffer("time_freqs", torch.exp(torch.arange(half, dtype=torch.float32) * (-math.log(10000.0) / (half - 1))),
                             persistent=False)
        use_sm = text_module == "scoremap"
        self.CLIP_ScoreMapModule = CLIP_ScoreMapModule if use_sm else None
        # if_MultiScoreMap False (reference models/drift_noise_model.py:113-114,130-131: ONE default ScoreMapModule() handed to create_net;
        # frozen spec): the module sits on the full-resolution level, its score map is embedded into score_map_chan
        # channels of that level's skip and is the one score map returned; the other levels carry no score-map channels
        self.n_sm = (self.depth if if_MultiScoreMap else 1) if use_sm else 0
        self.downs = nn.ModuleList()
        self.ups = nn.ModuleList()
        self.sm_embed = nn.ModuleList()
        self.level_dims = []
        for i in range(self.depth):
            din, dout = nf * mult[i], nf * mult[i + 1]
            smc = (score_map_ngf * score_map_ch_mult[i] if if_MultiScoreMap else score_map_chan) if i < self.n_sm else 0
            self.level_dims.append((din, dout, smc))
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
        self._ctx_cache = None
        self._idx_cache = {}

    # ---- helpers ---------------------------------------------------------------------------------
    def score_map_modules(self):
        """the net's ScoreMapModules as a list: one per level (if_MultiScoreMap) or the single module of level 0"""
        m = self.CLIP_ScoreMapModule
        if m is None:
            return []
        return list(m) if isinstance(m, (nn.ModuleList, list, tuple)) else [m]

    def resblocks(self):
        rbs = []
        for lv in self.downs:
            rbs += [lv.res1, lv.res2]
        rbs += [self.mid_res1, self.mid_res2]
        for up in self.ups:
            rbs += [up.res1, up.res2]
        rbs.append(self.final_res)
        return rbs

    def cross_attns(self):
        if not self.use_image_context:
            return []
        cas = []
        for lv in self.downs:
            cas += [lv.ca1, lv.ca2]
        cas.append(self.mid_ca)
        for up in self.ups:
            cas += [up.ca1, up.ca2]
        return cas

    def class_index(self, names, device):
        key = (tuple(names), str(device))
        hit = self._idx_cache.get(key)
        if hit is None:
            hit = torch.tensor([self.type_map_ind[n] for n in names], dtype=torch.int32, device=device)
            self._idx_cache[key] = hit
        return hit

    def _films(self, temb):
        """all ResBlock time projections Linear(SiLU(temb)) in ONE launch -> per-block [B, 2C] views."""
        rbs = self.resblocks()
        w, b = _PREP.get(("film", self), [p for rb in rbs for p in (rb.mlp.weight, rb.mlp.bias)],
                         lambda: (torch.cat([rb.mlp.weight for rb in rbs], 0).detach().contiguous(),
                                  torch.cat([rb.mlp.bias for rb in rbs], 0).detach().contiguous()))
        allf = ops.linear(temb, w, b, act_in=ops.ACT_SILU)
        out, o = {}, 0
        for rb in rbs:
            out[id(rb)] = allf[:, o:o + 2 * rb.dim_out]
            o += 2 * rb.dim_out
        return out

    def _ctx_vecs(self, ctx):
        """single-token image context: per-block vectors, cached across denoising steps (ctx is constant)."""
        cas = self.cross_attns()
        key = (ctx._version, tuple(ctx.shape), _weight_epoch()) + tuple((p.data_ptr(), p._version) for ca in cas
                                                        for p in (ca.v_proj.weight, ca.proj.weight, ca.proj.bias))
        c = self._ctx_cache
        if c is not None and c[0] == key and c[2]() is ctx:
            return c[1]
        vecs = {id(ca): ca.single_token_vec(ctx) for ca in cas}
        self._ctx_cache = (key, vecs, weakref.ref(ctx))
        return vecs

    # ---- forward ---------------------------------------------------------------------------------
    def forward(self, x_a, x_b, t, names, text_encoder, image_context=None):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .unet_autograd import forward_train  # hand-written backward kernels
            return forward_train(self, x_a, x_b, t, names, text_encoder, image_context)
        return self.forward_infer(x_a, x_b, t, names, text_encoder, image_context)

    @torch.no_grad()
    def forward_infer(self, x_a, x_b, t, names, text_encoder, image_context=None):
        dev = x_a.device
        B, _, H, W = x_a.shape
        if not torch.is_tensor(t):
            t = torch.full((B,), float(t), dtype=torch.float32, device=dev)
        t = t.reshape(-1).to(device=dev, dtype=torch.float32)
        if t.numel() == 1 and B > 1:
            t = t.expand(B)
        t = t.contiguous()
        idx = self.class_index(names, dev)
        ctx = image_context if self.use_image_context else None
        single = ctx is not None and ctx.shape[1] == 1
        general = ctx is not None and not single
        if ctx is not None:
            ctx = ctx.contiguous()
        vecs = self._ctx_vecs(ctx) if single else {}

        def ca_vec(ca_name, holder):
            return vecs[id(getattr(holder, ca_name))] if single else None

        # sinusoidal embedding + both linears of the time MLP: one launch where the MLP has the kernel's 64 -> 256 -> 256 geometry
        if self.nf == 64 and self.time_dim == 256:
            temb = ops.time_mlp(t, self.time_freqs, self.time_mlp[0].weight, self.time_mlp[0].bias, self.time_mlp[2].weight, self.time_mlp[2].bias)
        else:
            temb = ops.linear(ops.linear(ops.time_embed(t, self.nf, self.time_freqs), self.time_mlp[0].weight, self.time_mlp[0].bias, act_out=ops.ACT_GELU),
                              self.time_mlp[2].weight, self.time_mlp[2].bias)
        films = self._films(temb)

        x = ops.conv2d(x_a, packed(self.init_conv), self.init_conv.bias, 7, self.nf, src1=x_b)
        x_ = x
        hs, sms = [], []
        use_sm = self.CLIP_ScoreMapModule is not None
        sm_feats, sm_skips = [], []
        for i, lv in enumerate(self.downs):
            din, dout, smc = self.level_dims[i]
            Hi, Wi = x.shape[2], x.shape[3]
            x = lv.res1.run(x, None, films[id(lv.res1)], vec=ca_vec("ca1", lv))
            if general:
                x = lv.ca1.run(x, ctx)
            hs.append(x)
            use_sm = i < self.n_sm
            if use_sm:
                skip = torch.empty((B, din + smc, Hi, Wi), device=dev, dtype=torch.float32)
                xo = skip[:, :din]
            else:
                skip, xo = None, None
            x = lv.res2.run(x, None, films[id(lv.res2)], vec=ca_vec("ca2", lv), out=None if general else xo)
            if general:
                x = lv.ca2.run(x, ctx, out=xo)
            if use_sm:
                sm_feats.append(x)      # = skip[:, :din]; the score-map channels skip[:, din:] are filled below
                sm_skips.append(skip)
                hs.append(skip)
            else:
                hs.append(x)
            x = lv.down.run(x)
        side = cur = None
        if self.n_sm > 0:
            # A level's ScoreMapModule only feeds its skip connection (and the returned score maps), which the decoder reads much
            # later: the four modules run here, together -- every latency-bound token-side launch of their decoder chains serves all
            # four levels at once (decoder_tokens_grouped), 23 launches per net instead of 92.
            def smm_phase():
                smms = self.score_map_modules()
                texts = [m.text_embeddings(text_encoder, B) for m in smms]
                tvs = decoder_tokens_grouped(smms, sm_feats, texts)
                scores = _scoremaps(sm_feats, [tv.reshape(B, m.n_cls, feat.shape[1]) for m, feat, tv in zip(smms, sm_feats, tvs)], idx)
                for i, ((score, sel), skip) in enumerate(zip(scores, sm_skips)):
                    din, dout, smc = self.level_dims[i]
                    ops.conv2d(score, packed(self.sm_embed[i]), self.sm_embed[i].bias, 3, smc, out=skip[:, din:])
                    sms.append(sel)
            if SMM_SIDE and x.is_cuda:
                if getattr(self, "_side_stream", None) is None:
                    self._side_stream = torch.cuda.Stream()
                side, cur = self._side_stream, torch.cuda.current_stream()
                side.wait_stream(cur)            # fork: the encoder's features are complete on the net's stream
                with torch.cuda.stream(side):
                    smm_phase()
                for sel in sms:
                    sel.record_stream(cur)
            else:
                smm_phase()
        x = self.mid_res1.run(x, None, films[id(self.mid_res1)])
        x = self.mid_attn.run(x, vec=ca_vec("mid_ca", self))
        if general:
            x = self.mid_ca.run(x, ctx)
        x = self.mid_res2.run(x, None, films[id(self.mid_res2)])
        if side is not None:
            side.wait_stream(cur)                # the side stream (skips filled) waits for the mid blocks and carries on with the decoder
            for t in [x, x_] + hs:
                t.record_stream(side)
            with torch.cuda.stream(side):
                pred = self._decode(x, x_, hs, films, vecs if single else None, general, ctx, idx)
            pred.record_stream(cur)
            return (pred, sms) if self.text_module == "scoremap" else pred
        return self._decode(x, x_, hs, films, vecs if single else None, general, ctx, idx, sms if self.text_module == "scoremap" else None)

    def _decode(self, x, x_, hs, films, vecs, general, ctx, idx, sms=None):
        def ca_vec(ca_name, holder):
            return vecs[id(getattr(holder, ca_name))] if vecs is not None else None
        for up in self.ups:
            x = up.res1.run(x, hs.pop(), films[id(up.res1)], vec=ca_vec("ca1", up))
            if general:
                x = up.ca1.run(x, ctx)
            x = up.res2.run(x, hs.pop(), films[id(up.res2)], vec=ca_vec("ca2", up))
            if general:
                x = up.ca2.run(x, ctx)
            x = up.up.run(x)
        x = self.final_res.run(x, x_, films[id(self.final_res)])
        # final 3x3 conv to out_nc channels + per-sample class pick, computed as ONE channel per sample
        pred = ops.conv3x3_select(x, self.final_conv.weight.detach(), self.final_conv.bias, idx)
        if self.text_module == "scoremap" and sms is not None:
            return pred, sms
        return pred
