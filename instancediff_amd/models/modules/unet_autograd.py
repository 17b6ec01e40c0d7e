"""Training-mode forward of `LearnableForwardUNet_MultiScoreMap`: the same network as forward_infer, composed of
autograd Functions (instancediff_amd/train_ops.py) whose forward and backward are the HIP kernels.  ResBlocks keep
the inference fusion (one Function, hand-written backward); attention / ScoreMapModule chains are expressed with
batched-GEMM + softmax + LayerNorm Functions so that autograd derives their backward from those primitives."""
import torch

from ... import ops
from .MSM_degEmb_Unet import branch_gains
from ...train_ops import (ActFn, AddFn, AddVecFn, BLayerNormFn, BLinear3Fn, BLinearFn, BgemmFn, ChanLayerNormFn, ChanNormalizeFn, CompactMemFn,
                          ConvFn, GatherChannelFn, HeadFoldFn, HeadFoldInLFn, HeadFoldOutLFn, JoinFn, LayerNormRowsFn, Linear3Fn, LinearFn,
                          ResBlockFn, ScaleColsFn, SkipCatFn, SmmXattnFn, SoftmaxRowsFn, StackFn, StackParamsFn, StackRowsFn, TokenAttnFn,
                          UnstackFn, _Slot, dropout, fork)


# IDIFF_FUSED_XATTN=0: the ScoreMapModule cross-attention of the training path as batched GEMMs + softmax (autograd-derived backward)
import os  # noqa: E402
FUSED_XATTN = bool(int(os.environ.get("IDIFF_FUSED_XATTN", "1")))
# IDIFF_TRAIN_COMPACT=0: the 64-channel levels' ScoreMapModules attend to the materialised 256-row memory in the training step too (r04)
TRAIN_COMPACT = bool(int(os.environ.get("IDIFF_TRAIN_COMPACT", "1")))
# IDIFF_TRAIN_STACKED=0: every ScoreMapModule's decoder chain by itself, right behind its level (r04 .. first half of r05)
TRAIN_STACKED = bool(int(os.environ.get("IDIFF_TRAIN_STACKED", "1")))
_CEN = {}


def _centering(n, dev):
    """I - 11^T / n (symmetric): centring over the n outputs of the memory Linear as a matrix product"""
    key = (n, str(dev))
    if key not in _CEN:
        _CEN[key] = (torch.eye(n, dtype=torch.float64) - 1.0 / n).float().to(dev)
    return _CEN[key]


def _memory_fold(mp, Cm, uses):
    """The weight-side algebra of the compact memory (the sampling path does it once per weight version on the host,
    MSM_degEmb_Unet._fold_memory_affine) as differentiable library launches, every step:
        Wc = Cen W, bc = Cen b (centred over the Wd outputs);  Pt = g2 (.) [Wc^T ; bc ; 0]  [Cm, Wd]  (P^T: LayerNorm_Wd(W xh + b) = P m + b2)
        gram = Wc^T Wc / Wd,  hvec = Wc^T bc / Wd,  evar = |bc|^2 / Wd                      (the variance's quadratic form)
    -> (`uses` handles on Pt, gram [C,C], hvec [C], evar [1,1,1])"""
    lin, ln2 = mp[1], mp[2]
    Wd, C = lin.weight.shape
    cen = _centering(Wd, lin.weight.device)
    Wct = BgemmFn.apply(lin.weight[None], cen[None], True, False).reshape(C, Wd)   # W^T Cen = (Cen W)^T (reshape, not [0]: a view both ways)
    bc = LinearFn.apply(lin.bias[None, :], cen, None)                   # b Cen              [1, Wd]
    w0, w1, w2, w3 = fork(Wct, 4)
    c0, c1, c2, c3 = fork(bc, 4)
    Pt = ScaleColsFn.apply(StackRowsFn.apply(w0, c0, Cm), ln2.weight)
    gram = BgemmFn.apply(w1[None], w2[None], False, True, 1.0 / Wd).reshape(C, C)
    hvec = BgemmFn.apply(c1[None], w3[None], False, True, 1.0 / Wd).reshape(C)
    evar = BgemmFn.apply(c2[None], c3[None], False, True, 1.0 / Wd)
    return fork(Pt, uses), gram, hvec, evar


def _resblock(rb, src0, src1, temb_act, vec=None, out=None):
    """out: a preallocated destination (a channel slice of a skip buffer) the block's output is written into"""
    import torch.nn as nn
    film = LinearFn.apply(temb_act, rb.mlp.weight, rb.mlp.bias)
    ident = isinstance(rb.res_conv, nn.Identity)
    return ResBlockFn.apply(src0, src1, film, vec, rb.conv1.weight, rb.conv1.bias, rb.norm1.weight, rb.norm1.bias, rb.conv2.weight,
                            rb.conv2.bias, rb.norm2.weight, rb.norm2.bias, None if ident else rb.res_conv.weight,
                            None if ident else rb.res_conv.bias, rb.groups, rb.norm1.eps, None if out is None else _Slot(out))


def _heads_attention(q3, k3, v3, heads, scale):
    """token-major attention from batched GEMMs: q3 [B,Nq,C], k3/v3 [B,M,C] -> [B,Nq,C].  All heads in one batch ([B*heads, ., dh]
    through two permutes) instead of a slice per head: autograd's backward of a slice is a zero-filled full-size tensor plus an add
    per head -- hundreds of tiny ATen launches per step."""
    B, Nq, C = q3.shape
    M = k3.shape[1]
    dh = C // heads

    def split(x, n):
        return x.reshape(B, n, heads, dh).permute(0, 2, 1, 3).reshape(B * heads, n, dh)
    s = BgemmFn.apply(split(q3, Nq), split(k3, M), False, True)     # [B*heads, Nq, M]
    p = SoftmaxRowsFn.apply(s, scale)
    o = BgemmFn.apply(p, split(v3, M), False, False)                  # [B*heads, Nq, dh]
    return o.reshape(B, heads, Nq, dh).permute(0, 2, 1, 3).reshape(B, Nq, C)


def _ca_vec(ca, ctx):
    """single-token image context: proj(v_proj(ctx)) as a per-(sample, channel) vector"""
    B = ctx.shape[0]
    v = LinearFn.apply(ctx.reshape(B, -1), ca.v_proj.weight, None)
    return LinearFn.apply(v, ca.proj.weight.reshape(ca.dim, ca.dim), ca.proj.bias)


def _ca_general(ca, x, ctx):
    """x + proj(attention(q_proj(norm(x)), k_proj(ctx), v_proj(ctx))) for M > 1 context tokens (channel-major maps)"""
    B, C, H, W = x.shape
    M, N, heads = ctx.shape[1], H * W, ca.num_heads
    dh = C // heads
    xn = ChanLayerNormFn.apply(x, ca.norm.weight, ca.norm.bias, 1e-5)
    q = ConvFn.apply(xn, None, ca.q_proj.weight, None, 1, ops.CONV_NORMAL).reshape(B, C, N)
    c2 = ctx.reshape(B * M, -1)
    k = LinearFn.apply(c2, ca.k_proj.weight, None).reshape(B, M, C)
    v = LinearFn.apply(c2, ca.v_proj.weight, None).reshape(B, M, C)
    outs = []
    for h in range(heads):
        sl = slice(h * dh, (h + 1) * dh)
        s = BgemmFn.apply(q[:, sl, :], k[:, :, sl], True, True)        # q_h stored [dh][N] -> [N,dh] . [dh,M]
        p = SoftmaxRowsFn.apply(s, ca.scale)                             # [B,N,M]
        outs.append(BgemmFn.apply(v[:, :, sl], p, True, True))           # [dh,M] . [M,N] -> [B,dh,N]
    o = torch.cat(outs, dim=1).reshape(B, C, H, W)
    return AddFn.apply(x, ConvFn.apply(o, None, ca.proj.weight, ca.proj.bias, 1, ops.CONV_NORMAL), 1.0)


def _self_attention(sa, x, vec=None):
    B, C, H, W = x.shape
    N, heads = H * W, sa.num_heads
    dh = C // heads
    x, xr = fork(x, 2)
    xn = ChanLayerNormFn.apply(x, sa.norm.weight, sa.norm.bias, 1e-5)
    qkv = ConvFn.apply(xn, None, sa.qkv.weight, None, 1, ops.CONV_NORMAL).reshape(B, 3, heads, dh, N)
    q = qkv[:, 0].reshape(B * heads, dh, N)
    k = qkv[:, 1].reshape(B * heads, dh, N)
    v = qkv[:, 2].reshape(B * heads, dh, N)
    s = BgemmFn.apply(q, k, True, False)            # [N_q, dh] . [dh, N_k]
    p = SoftmaxRowsFn.apply(s, sa.scale)            # [BH, N_q, N_k]
    o = BgemmFn.apply(v, p, False, True)            # [dh, N_k] . [N_k, N_q] -> channel-major [BH, dh, N_q]
    o = o.reshape(B, C, H, W)
    y = AddFn.apply(xr, ConvFn.apply(o, None, sa.proj.weight, sa.proj.bias, 1, ops.CONV_NORMAL), 1.0)
    return AddVecFn.apply(y, vec) if vec is not None else y


def _smm(smm, feat, text_encoder, idx, feat_n=None):
    """ScoreMapModule forward in Function form -> (score [B,K,h,w], sel [B,1,h,w]).  feat_n: a second handle on the same feature map
    for the score map's normalisation (the caller's fork: the module reads the map twice)"""
    B, C, H, W = feat.shape
    K, N = smm.n_cls, H * W
    dec = smm.context_decoder
    Wd, heads = dec.width, dec.heads
    dh = Wd // heads
    text = smm.text_embeddings(text_encoder, B)
    t2d = text.reshape(B * K, smm.text_dim)
    t2d, t2d_v = fork(t2d, 2)  # text projection and text_to_visual
    mp = dec.memory_proj
    if feat_n is None:
        feat, feat_n = fork(feat, 2)  # consumed by the memory projection and by the score map's normalisation
    fused_x = FUSED_XATTN and heads * K <= 32 and Wd == 256 and N % 4 == 0
    compact = TRAIN_COMPACT and fused_x and C == 64 and feat.is_cuda
    if compact:
        # the (C + 1)-row pre-image of the memory (72 rows instead of 256) and the weights that fold its affine map into the attention
        Cm = 72
        pts, gram, hvec, evar = _memory_fold(mp, Cm, 2 * len(dec.decoder))
        pts = list(pts)
        b2rows = list(fork(mp[2].bias[None, :], len(dec.decoder)))  # a parameter with several consumers is forked too (no ATen accumulate)
        mem = CompactMemFn.apply(feat, mp[0].weight, mp[0].bias, gram, hvec, evar, Cm, mp[0].eps, mp[2].eps)
    else:
        fn = ChanLayerNormFn.apply(feat, mp[0].weight, mp[0].bias, mp[0].eps)
        m1 = ConvFn.apply(fn, None, mp[1].weight.reshape(Wd, C, 1, 1), mp[1].bias, 1, ops.CONV_NORMAL)
        mem = ChanLayerNormFn.apply(m1, mp[2].weight, mp[2].bias, mp[2].eps).reshape(B, Wd, N)
    tp = dec.text_proj
    x = LinearFn.apply(LayerNormRowsFn.apply(t2d, tp[0].weight, tp[0].bias, tp[0].eps), tp[1].weight, tp[1].bias)
    R = B * K
    # training-mode dropout of the decoder blocks (TransformerDecoderLayer(dropout=0.1): Attention.proj_drop on both attentions, the
    # MLP's inner Dropout and the block's output Dropout, models/_modified_BiomedCLIP.py:448-478,520-549); identity in eval()
    pd, tr = dec.dropout, smm.training
    mem_grad = {}  # the layers' gradients w.r.t. the shared memory are summed inside the fused backward kernel (SmmXattnFn)
    fused_t = K <= 8 and dh <= 64  # few-token self-attention: one fused forward / backward launch each

    def branch(y, g):  # TransformerDecoderLayer_scaled's per-channel gain on a residual branch (:586-589); plain layers: none
        return y if g is None else ScaleColsFn.apply(y, g)

    # Token rows r = (sample, class token).  No tensor is permuted or copied between the launches: the head dimension of the folded k / v
    # projections lives in the strides of the batched GEMMs (HeadFoldFn: rows of a sample ordered (token, head) -- the attention over
    # the memory treats its query rows independently), the three self-attention projections share one packed buffer, and a state
    # with two consumers (residual + LayerNorm) is forked, so its two gradients meet in one library launch.
    for layer in dec.decoder:
        sa, ca = layer.self_attn, layer.cross_attn
        g_sa, g_ca, g_mlp = branch_gains(layer)
        x, xr = fork(x, 2)
        n1 = LayerNormRowsFn.apply(x, layer.norm1.weight, layer.norm1.bias, layer.norm1.eps)
        if fused_t:
            a = TokenAttnFn.apply(Linear3Fn.apply(n1, sa.q_proj.weight, sa.k_proj.weight, sa.v_proj.weight), B, K, heads, sa.scale)
        else:
            n1a, n1b, n1c = fork(n1, 3)
            q = LinearFn.apply(n1a, sa.q_proj.weight, None).reshape(B, K, Wd)
            k = LinearFn.apply(n1b, sa.k_proj.weight, None).reshape(B, K, Wd)
            v = LinearFn.apply(n1c, sa.v_proj.weight, None).reshape(B, K, Wd)
            a = _heads_attention(q, k, v, heads, sa.scale).reshape(R, Wd)
        x = AddFn.apply(xr, branch(dropout(LinearFn.apply(a, sa.proj.weight, sa.proj.bias), pd, tr), g_sa), 1.0)
        x, xr = fork(x, 2)
        n2 = LayerNormRowsFn.apply(x, layer.norm2.weight, layer.norm2.bias, layer.norm2.eps)
        qc = LinearFn.apply(n2, ca.q_proj.weight, None)  # [R, Wd]
        # k/v projections folded onto the queries: qf_h = q_h Wk_h ; S = qf mem ; o = P mem^T ; av_h = o_h Wv_h^T.
        # All heads' query rows are stacked ([B, K*heads, Wd]) so `mem` is read once per product, not once per head.
        if compact:
            # mem = P m + b2: fold P into the k / v weights (S = q Wk P m up to a per-row constant; o_256 = P (sum_n p_n m_n) + b2)
            wv_a, wv_b = fork(ca.v_proj.weight, 2)
            wp_a, wp_b = fork(ca.proj.weight, 2)
            wkf = BgemmFn.apply(ca.k_proj.weight[None], pts.pop()[None], False, True).reshape(Wd, Cm)   # Wk P
            wvp = BgemmFn.apply(wv_a[None], pts.pop()[None], False, True).reshape(Wd, Cm)               # Wv P
            bvf = LinearFn.apply(b2rows.pop(), wv_b, None)                                 # b2 Wv^T [1, Wd]: the softmax weights sum to 1
            pbias = LinearFn.apply(bvf, wp_a, ca.proj.bias).reshape(Wd)                    # ... carried through the output projection
            qf = HeadFoldFn.apply(qc, wkf, heads, "in").reshape(B, K * heads, Cm)          # row = k*heads + h
            o = SmmXattnFn.apply(qf, mem, ca.scale, mem_grad)
            av = HeadFoldFn.apply(o.reshape(R, heads, Cm), wvp, heads, "out")              # [R, Wd]
            x = AddFn.apply(xr, branch(dropout(LinearFn.apply(av, wp_b, pbias), pd, tr), g_ca), 1.0)
        else:
            qf = HeadFoldFn.apply(qc, ca.k_proj.weight, heads, "in").reshape(B, K * heads, Wd)   # row = k*heads + h
            if fused_x:
                o = SmmXattnFn.apply(qf, mem, ca.scale, mem_grad)        # one fused forward, one fused backward pass over the keys
            else:
                s_ = BgemmFn.apply(qf, mem, False, False)                # [B, K*heads, N]
                p = SoftmaxRowsFn.apply(s_, ca.scale)
                o = BgemmFn.apply(p, mem, False, True)                   # [B, K*heads, Wd]
            av = HeadFoldFn.apply(o.reshape(R, heads, Wd), ca.v_proj.weight, heads, "out")       # [R, Wd]
            x = AddFn.apply(xr, branch(dropout(LinearFn.apply(av, ca.proj.weight, ca.proj.bias), pd, tr), g_ca), 1.0)
        x, xr = fork(x, 2)
        n3 = LayerNormRowsFn.apply(x, layer.norm3.weight, layer.norm3.bias, layer.norm3.eps)
        hm = dropout(ActFn.apply(LinearFn.apply(n3, layer.mlp[0].weight, layer.mlp[0].bias), ops.ACT_GELU), pd, tr)
        x = AddFn.apply(xr, branch(dropout(LinearFn.apply(hm, layer.mlp[3].weight, layer.mlp[3].bias), pd, tr), g_mlp), 1.0)
    op = dec.out_proj
    diff = LinearFn.apply(LayerNormRowsFn.apply(x, op[0].weight, op[0].bias, op[0].eps), op[1].weight, op[1].bias)  # [R, C]
    t2v = LinearFn.apply(t2d_v, smm.text_to_visual.weight, smm.text_to_visual.bias)
    tv = AddFn.apply(t2v, ScaleColsFn.apply(diff, smm.gamma), 1.0)
    tvn = ChanNormalizeFn.apply(tv.reshape(R, C, 1)).reshape(B, K, C)
    fnm = ChanNormalizeFn.apply(feat_n).reshape(B, C, N)
    score = BgemmFn.apply(tvn, fnm, False, False).reshape(B, K, H, W)
    score, score_g = fork(score, 2)  # the embedding conv of the skip and the class pick of the pyramid loss
    sel = GatherChannelFn.apply(score_g, idx)
    return score, sel


def _skip_with_embedding(net, i, score, x_skip, skipbuf, din):
    """the level's skip cat(x, conv3x3(score map)): written in place into the skip buffer where there is one"""
    if skipbuf is not None:
        emb = ConvFn.apply(score, None, net.sm_embed[i].weight, net.sm_embed[i].bias, 3, ops.CONV_NORMAL, _Slot(skipbuf[:, din:].detach()))
        return SkipCatFn.apply(x_skip, emb, _Slot(skipbuf))
    emb = ConvFn.apply(score, None, net.sm_embed[i].weight, net.sm_embed[i].bias, 3, ops.CONV_NORMAL)
    return ("cat", x_skip, emb)


def _stackable(smms):
    """the stacked form covers plain ContextDecoders of one geometry (the default model); anything else runs level by level"""
    if len(smms) < 2:
        return False
    d0 = smms[0].context_decoder
    for m in smms:
        d = m.context_decoder
        if type(d).__name__ != "ContextDecoder" or (d.width, d.heads, len(d.decoder), d.dropout) != (d0.width, d0.heads, len(d0.decoder), d0.dropout):
            return False
        if (m.n_cls, m.text_dim, m.training) != (smms[0].n_cls, smms[0].text_dim, smms[0].training) or d.width // d.heads > 64 or m.n_cls > 8:
            return False
    return True


def _smm_all(smms, feats, feats_n, text_encoder, idx):
    """The ScoreMapModules of a net's L levels with their decoder token chains STACKED (r05): every token-side operation -- the same
    [B*K, 256]-row LayerNorm / linear / attention of each level, with its own weights -- is ONE batch-L launch forward and backward
    (weights gathered into [L, ..] stacks by one launch per step, StackParamsFn; BLinearFn / BLinear3Fn / BLayerNormFn; the five-token
    self-attention over L*B samples), as the sampling path has advanced the four chains in lock step since r03.  Per level remain what
    sees the image: the memory (compact for 64 channels), the fold of its affine map, the cross-attention over the memory, and the
    head of the module (out_proj / text_to_visual / score map: the channel count differs).  -> [(score, sel)] per level"""
    L = len(smms)
    decs = [m.context_decoder for m in smms]
    Wd, heads, nlayers = decs[0].width, decs[0].heads, len(decs[0].decoder)
    dh = Wd // heads
    B = feats[0].shape[0]
    K = smms[0].n_cls
    R = B * K
    pd, tr = decs[0].dropout, smms[0].training
    dev = feats[0].device
    # ---- per level: memory (+ fold) -------------------------------------------------------------------------------------------
    lv = []
    for m, dec, feat in zip(smms, decs, feats):
        Bc, C, H, W = feat.shape
        N = H * W
        mp = dec.memory_proj
        fused_x = FUSED_XATTN and heads * K <= 32 and Wd == 256 and N % 4 == 0
        compact = TRAIN_COMPACT and fused_x and C == 64
        e = dict(C=C, N=N, H=H, W=W, fused_x=fused_x, compact=compact, Cm=72 if compact else Wd, mem_grad={}, mp=mp)
        if compact:
            pts, gram, hvec, evar = _memory_fold(mp, 72, 2 * nlayers)
            e["pts"] = list(pts)
            e["b2rows"] = list(fork(mp[2].bias[None, :], nlayers))
            e["mem"] = CompactMemFn.apply(feat, mp[0].weight, mp[0].bias, gram, hvec, evar, 72, mp[0].eps, mp[2].eps)
        else:
            fn = ChanLayerNormFn.apply(feat, mp[0].weight, mp[0].bias, mp[0].eps)
            m1 = ConvFn.apply(fn, None, mp[1].weight.reshape(Wd, C, 1, 1), mp[1].bias, 1, ops.CONV_NORMAL)
            e["mem"] = ChanLayerNormFn.apply(m1, mp[2].weight, mp[2].bias, mp[2].eps).reshape(B, Wd, N)
        lv.append(e)
    # ---- stacked parameters: one gather launch ---------------------------------------------------------------------------------
    kinds, flat = [], []

    def add(name, tensors):
        kinds.append(name)
        flat.extend(tensors)
    add("tp_g", [d.text_proj[0].weight for d in decs]), add("tp_b", [d.text_proj[0].bias for d in decs])
    add("tp_w", [d.text_proj[1].weight for d in decs]), add("tp_wb", [d.text_proj[1].bias for d in decs])
    proj_w_rest = []  # per layer, per level: the handle on cross_attn.proj.weight kept for the compact levels' bias fold
    for li in range(nlayers):
        lay = [d.decoder[li] for d in decs]
        add(f"{li}.n1g", [l.norm1.weight for l in lay]), add(f"{li}.n1b", [l.norm1.bias for l in lay])
        add(f"{li}.wq", [l.self_attn.q_proj.weight for l in lay]), add(f"{li}.wk", [l.self_attn.k_proj.weight for l in lay])
        add(f"{li}.wv", [l.self_attn.v_proj.weight for l in lay])
        add(f"{li}.wp", [l.self_attn.proj.weight for l in lay]), add(f"{li}.bp", [l.self_attn.proj.bias for l in lay])
        add(f"{li}.n2g", [l.norm2.weight for l in lay]), add(f"{li}.n2b", [l.norm2.bias for l in lay])
        add(f"{li}.wqc", [l.cross_attn.q_proj.weight for l in lay])
        hs_, rest = [], []
        for l, e in zip(lay, lv):
            if e["compact"]:
                a, b_ = fork(l.cross_attn.proj.weight, 2)
                hs_.append(a), rest.append(b_)
            else:
                hs_.append(l.cross_attn.proj.weight), rest.append(None)
        add(f"{li}.wpc", hs_)
        proj_w_rest.append(rest)
        add(f"{li}.n3g", [l.norm3.weight for l in lay]), add(f"{li}.n3b", [l.norm3.bias for l in lay])
        add(f"{li}.w0", [l.mlp[0].weight for l in lay]), add(f"{li}.b0", [l.mlp[0].bias for l in lay])
        add(f"{li}.w3", [l.mlp[3].weight for l in lay]), add(f"{li}.b3", [l.mlp[3].bias for l in lay])
    S = dict(zip(kinds, StackParamsFn.apply(L, len(kinds), *flat)))
    # ---- stacked chain -----------------------------------------------------------------------------------------------------------
    t2ds, t2vs_in = [], []
    for m in smms:
        t2d = m.text_embeddings(text_encoder, B).reshape(R, m.text_dim)
        a, b_ = fork(t2d, 2)  # text projection (stacked) and text_to_visual (per level)
        t2ds.append(a), t2vs_in.append(b_)
    X = BLinearFn.apply(BLayerNormFn.apply(StackFn.apply(*t2ds), S["tp_g"], S["tp_b"], decs[0].text_proj[0].eps), S["tp_w"], S["tp_wb"])
    for li in range(nlayers):
        lay = [d.decoder[li] for d in decs]
        sa0 = lay[0].self_attn
        X, Xr = fork(X, 2)
        n1 = BLayerNormFn.apply(X, S[f"{li}.n1g"], S[f"{li}.n1b"], lay[0].norm1.eps)
        qkv = BLinear3Fn.apply(n1, S[f"{li}.wq"], S[f"{li}.wk"], S[f"{li}.wv"])
        a = TokenAttnFn.apply(qkv.reshape(L * R, 3 * Wd), L * B, K, heads, sa0.scale).reshape(L, R, Wd)
        X = AddFn.apply(Xr, dropout(BLinearFn.apply(a, S[f"{li}.wp"], S[f"{li}.bp"]), pd, tr), 1.0)
        X, Xr = fork(X, 2)
        n2 = BLayerNormFn.apply(X, S[f"{li}.n2g"], S[f"{li}.n2b"], lay[0].norm2.eps)
        QC = BLinearFn.apply(n2, S[f"{li}.wqc"], None)
        with torch.no_grad():
            AV = torch.empty((L, R, Wd), device=dev, dtype=torch.float32)
        shared_q, avs, pbs = {}, [], []
        for l, (layer, e) in enumerate(zip(lay, lv)):
            ca = layer.cross_attn
            Cm = e["Cm"]
            if e["compact"]:
                wv_a, wv_b = fork(ca.v_proj.weight, 2)
                wkf = BgemmFn.apply(ca.k_proj.weight[None], e["pts"].pop()[None], False, True).reshape(Wd, Cm)   # Wk P
                wvp = BgemmFn.apply(wv_a[None], e["pts"].pop()[None], False, True).reshape(Wd, Cm)               # Wv P
                bvf = LinearFn.apply(e["b2rows"].pop(), wv_b, None)                                 # b2 Wv^T
                pbs.append(LinearFn.apply(bvf, proj_w_rest[li][l], ca.proj.bias))                   # ... through the output projection [1, Wd]
            else:
                wkf, wvp = ca.k_proj.weight, ca.v_proj.weight
                pbs.append(ca.proj.bias[None, :])
            qf = HeadFoldInLFn.apply(QC, wkf, heads, l, shared_q).reshape(B, K * heads, Cm)        # row = k*heads + h
            if e["fused_x"]:
                o = SmmXattnFn.apply(qf, e["mem"], ca.scale, e["mem_grad"])
            else:
                s_ = BgemmFn.apply(qf, e["mem"], False, False)
                o = BgemmFn.apply(SoftmaxRowsFn.apply(s_, ca.scale), e["mem"], False, True)
            avs.append(HeadFoldOutLFn.apply(o.reshape(R, heads, Cm), wvp, heads, _Slot(AV[l].detach())))
        AVs = JoinFn.apply(_Slot(AV), *avs)
        PB = StackFn.apply(*pbs).reshape(L, Wd)
        X = AddFn.apply(Xr, dropout(BLinearFn.apply(AVs, S[f"{li}.wpc"], PB), pd, tr), 1.0)
        X, Xr = fork(X, 2)
        n3 = BLayerNormFn.apply(X, S[f"{li}.n3g"], S[f"{li}.n3b"], lay[0].norm3.eps)
        hm = dropout(ActFn.apply(BLinearFn.apply(n3, S[f"{li}.w0"], S[f"{li}.b0"]), ops.ACT_GELU), pd, tr)
        X = AddFn.apply(Xr, dropout(BLinearFn.apply(hm, S[f"{li}.w3"], S[f"{li}.b3"]), pd, tr), 1.0)
    xs = UnstackFn.apply(X)
    # ---- per level: the module's head ------------------------------------------------------------------------------------------
    outs = []
    for m, dec, e, x, t2d, fn_ in zip(smms, decs, lv, xs, t2vs_in, feats_n):
        C, H, W, N = e["C"], e["H"], e["W"], e["N"]
        op = dec.out_proj
        diff = LinearFn.apply(LayerNormRowsFn.apply(x, op[0].weight, op[0].bias, op[0].eps), op[1].weight, op[1].bias)  # [R, C]
        t2v = LinearFn.apply(t2d, m.text_to_visual.weight, m.text_to_visual.bias)
        tv = AddFn.apply(t2v, ScaleColsFn.apply(diff, m.gamma), 1.0)
        tvn = ChanNormalizeFn.apply(tv.reshape(R, C, 1)).reshape(B, K, C)
        fnm = ChanNormalizeFn.apply(fn_).reshape(B, C, N)
        score = BgemmFn.apply(tvn, fnm, False, False).reshape(B, K, H, W)
        score, score_g = fork(score, 2)
        outs.append((score, GatherChannelFn.apply(score_g, idx)))
    return outs


def forward_train(net, x_a, x_b, t, names, text_encoder, image_context=None):
    if any(m.context_decoder.if_flash for m in net.score_map_modules()):
        raise RuntimeError("score_map_if_flash (the fp16 form of the ScoreMapModule decoder attentions) is an inference-only variant: "
                           "no backward is built for it")
    dev = x_a.device
    B, _, H, W = x_a.shape
    if not torch.is_tensor(t):
        t = torch.full((B,), float(t), dtype=torch.float32, device=dev)
    t = t.reshape(-1).to(device=dev, dtype=torch.float32)
    if t.numel() == 1 and B > 1:
        t = t.expand(B)
    t = t.contiguous()
    idx = net.class_index(names, dev)
    ctx = image_context.contiguous() if (net.use_image_context and image_context is not None) else None
    single = ctx is not None and ctx.shape[1] == 1
    general = ctx is not None and not single

    def vec_of(holder, name):
        return _ca_vec(getattr(holder, name), ctx) if single else None

    temb0 = ops.time_embed(t, net.nf, net.time_freqs)
    hmid = ActFn.apply(LinearFn.apply(temb0, net.time_mlp[0].weight, net.time_mlp[0].bias), ops.ACT_GELU)
    temb = LinearFn.apply(hmid, net.time_mlp[2].weight, net.time_mlp[2].bias)
    tact = ActFn.apply(temb, ops.ACT_SILU)
    tacts = list(fork(tact, len(net.resblocks())))  # one handle per ResBlock's time projection: their gradients meet in library launches

    # A feature map with several consumers is handed out through fork(): the consumers' gradients then meet in ONE library launch
    # (idiff_sum_n) instead of autograd's pairwise ATen adds.  A level's skip cat(x, score-map embedding) is never copied together: both
    # producers write into the channel slices of one buffer (SkipCatFn), as in the sampling path.
    x = ConvFn.apply(x_a.contiguous(), x_b.contiguous(), net.init_conv.weight, net.init_conv.bias, 7, ops.CONV_NORMAL)
    x, x_ = fork(x, 2)
    hs, sms = [], []
    smms = net.score_map_modules()   # one per level, or the single module of level 0 (if_MultiScoreMap False)
    stacked = bool(smms) and TRAIN_STACKED and x.is_cuda and _stackable(smms)
    pending = []
    for i, lv in enumerate(net.downs):
        use_sm = i < net.n_sm
        x = _resblock(lv.res1, x, None, tacts.pop(), vec_of(lv, "ca1"))
        if general:
            x = _ca_general(lv.ca1, x, ctx)
        x, xs = fork(x, 2)
        hs.append(xs)
        din, dout, smc = net.level_dims[i]
        skipbuf = None
        if use_sm and not general:
            with torch.no_grad():
                skipbuf = torch.empty((B, din + smc, x.shape[2], x.shape[3]), device=dev, dtype=torch.float32)
        x = _resblock(lv.res2, x, None, tacts.pop(), vec_of(lv, "ca2"), out=None if skipbuf is None else skipbuf[:, :din].detach())
        if general:
            x = _ca_general(lv.ca2, x, ctx)
        if use_sm:
            x, x_smm, x_smm2, x_skip = fork(x, 4)  # down conv, memory projection, score-map normalisation, skip
            if stacked:
                # the level's ScoreMapModule only feeds its skip (read by the decoder) and the pyramid loss: the L modules run together
                # behind the encoder, their token chains stacked (_smm_all)
                pending.append((len(hs), i, x_smm, x_smm2, x_skip, skipbuf))
                hs.append(None)
                sms.append(None)
            else:
                score, sel = _smm(smms[i], x_smm, text_encoder, idx, feat_n=x_smm2)
                sms.append(sel)
                hs.append(_skip_with_embedding(net, i, score, x_skip, skipbuf, din))
        else:
            x, xs = fork(x, 2)
            hs.append(xs)
        down = lv.down
        mode = ops.CONV_UNSHUFFLE2 if type(down).__name__ == "Downsample" else ops.CONV_NORMAL
        x = ConvFn.apply(x, None, down.conv.weight, down.conv.bias, 1 if mode == ops.CONV_UNSHUFFLE2 else 3, mode)
    if pending:
        res = _smm_all([smms[p_[1]] for p_ in pending], [p_[2] for p_ in pending], [p_[3] for p_ in pending], text_encoder, idx)
        for (slot_i, i, _, _, x_skip, skipbuf), (score, sel) in zip(pending, res):
            sms[i] = sel
            hs[slot_i] = _skip_with_embedding(net, i, score, x_skip, skipbuf, net.level_dims[i][0])
    x = _resblock(net.mid_res1, x, None, tacts.pop())
    x = _self_attention(net.mid_attn, x, vec_of(net, "mid_ca"))
    if general:
        x = _ca_general(net.mid_ca, x, ctx)
    x = _resblock(net.mid_res2, x, None, tacts.pop())
    for up in net.ups:
        skip = hs.pop()
        skip = torch.cat([skip[1], skip[2]], dim=1) if isinstance(skip, tuple) else skip
        x = _resblock(up.res1, x, skip, tacts.pop(), vec_of(up, "ca1"))
        if general:
            x = _ca_general(up.ca1, x, ctx)
        x = _resblock(up.res2, x, hs.pop(), tacts.pop(), vec_of(up, "ca2"))
        if general:
            x = _ca_general(up.ca2, x, ctx)
        u = up.up
        mode = ops.CONV_UPSAMPLE2 if type(u).__name__ == "Upsample" else ops.CONV_NORMAL
        x = ConvFn.apply(x, None, u.conv.weight, u.conv.bias, 3, mode)
    x = _resblock(net.final_res, x, x_, tacts.pop())
    if x.shape[3] % 4 == 0 and x.shape[1] <= 256:
        from ...train_ops import SelectConvFn
        pred = SelectConvFn.apply(x, net.final_conv.weight, net.final_conv.bias, idx)   # output conv + class pick, one channel per sample
    else:
        out = ConvFn.apply(x, None, net.final_conv.weight, net.final_conv.bias, 3, ops.CONV_NORMAL)
        pred = GatherChannelFn.apply(out, idx)
    if net.text_module == "scoremap":
        return pred, sms
    return pred
