"""Op-level parity: every C-ABI kernel vs a plain PyTorch CPU (fp64 where cheap) reference of the same op.
Runs on the GPU box only (-m gpu)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from instancediff_amd import ops  # noqa: E402
from oracle import philox_ref, sde_ref  # noqa: E402

DEV = "cuda"


def _close(got, ref, tol, what=""):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    scale = max(float(ref.abs().max()), 1e-6)
    err = float((got - ref).abs().max())
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (tol {tol})"


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _conv_tol():
    """fp32 error bound (of the output range) of the kernel that served the last conv2d: direct, F(2x2,3x3), F(4x4,3x3)."""
    return {0: 2e-6, 1: 6e-6, 3: 4e-5, 4: 4e-5, 5: 2e-6}[ops._lib.load().idiff_conv2d_last_algo()]


def silu64(x):
    return x / (1 + torch.exp(-x))


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,Cin,Cout,H,W,ks", [
    (2, 64, 64, 32, 32, 3),     # TW=32, vector weight path
    (1, 16, 128, 16, 16, 3),    # TW=16
    (2, 24, 64, 8, 8, 3),       # TW=8, tile bigger than the image
    (1, 8, 5, 40, 36, 3),       # ragged H/W, Cout < 64 (scalar weight path)
    (2, 64, 192, 32, 32, 1),    # 1x1
    (1, 40, 64, 16, 48, 1),     # 1x1, Cin not a multiple of the chunk
    (2, 2, 64, 32, 32, 7),      # 7x7 input layer
    (1, 1, 64, 24, 40, 7),
    (1, 128, 128, 64, 64, 3),
])
def test_conv_plain(B, Cin, Cout, H, W, ks):
    g = _g(1)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / math.sqrt(Cin * ks * ks)
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=ks // 2)
    wpk = ops.pack_conv_weight(w.to(DEV))
    out = ops.conv2d(x.to(DEV), wpk, b.to(DEV), ks, Cout)
    _close(out, ref, _conv_tol(), "conv")


def test_conv_concat_prologue_epilogue_stats():
    g = _g(2)
    B, C0, C1, Cout, H, W = 2, 32, 48, 64, 32, 32
    x0 = torch.randn(B, C0, H, W, generator=g)
    x1 = torch.randn(B, C1, H, W, generator=g)
    w = torch.randn(Cout, C0 + C1, 3, 3, generator=g) / math.sqrt((C0 + C1) * 9)
    b = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, H, W, generator=g)
    vec = torch.randn(B, Cout, generator=g)
    aux = torch.randn(B, Cout, H, W, generator=g)
    aa, ab = torch.randn(B, Cout, generator=g), torch.randn(B, Cout, generator=g)
    raw = F.conv2d(torch.cat([x0, x1], 1).double(), w.double(), b.double(), padding=1)
    ref = raw + res.double() + vec.double()[:, :, None, None] + silu64(aa.double()[:, :, None, None] * aux.double() + ab.double()[:, :, None, None])
    # src1 as a channel slice of a bigger buffer (batch stride > C1*H*W)
    big = torch.zeros(B, C1 + 16, H, W)
    big[:, :C1] = x1
    bigd = big.to(DEV)
    wpk = ops.pack_conv_weight(w.to(DEV))
    out, stats = ops.conv2d(x0.to(DEV), wpk, b.to(DEV), 3, Cout, src1=bigd[:, :C1], res=res.to(DEV), vec=vec.to(DEV),
                            aux=(aux.to(DEV), aa.to(DEV), ab.to(DEV)), want_stats=True)
    _close(out, ref, 2e-6, "conv concat+epilogue")
    # stats are of the raw conv output (acc + bias)
    s = stats.cpu().double().sum(dim=1)  # [B, Cout, 2]
    _close(s[..., 0], raw.sum(dim=(2, 3)), 1e-5, "stats sum")
    _close(s[..., 1], (raw ** 2).sum(dim=(2, 3)), 1e-5, "stats sumsq")
    # prologue (single source): silu(a*x+b) applied before zero padding
    pa, pb = torch.randn(B, C0, generator=g), torch.randn(B, C0, generator=g)
    w2 = torch.randn(Cout, C0, 3, 3, generator=g) / math.sqrt(C0 * 9)
    act = silu64(pa.double()[:, :, None, None] * x0.double() + pb.double()[:, :, None, None])
    ref2 = F.conv2d(act, w2.double(), None, padding=1)
    out2 = ops.conv2d(x0.to(DEV), ops.pack_conv_weight(w2.to(DEV)), None, 3, Cout, pro=(pa.to(DEV), pb.to(DEV)))
    _close(out2, ref2, 2e-6, "conv prologue")


@pytest.mark.parametrize("H,W", [(16, 16), (8, 8), (32, 64)])
def test_conv_upsample_and_unshuffle(H, W):
    g = _g(3)
    B, Cin, Cout = 2, 32, 64
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), w.double(), b.double(), padding=1)
    out = ops.conv2d(x.to(DEV), ops.pack_conv_weight(w.to(DEV)), b.to(DEV), 3, Cout, mode=ops.CONV_UPSAMPLE2)
    _close(out, ref, _conv_tol(), "upsample conv")
    w1 = torch.randn(Cout, Cin * 4, 1, 1, generator=g) / math.sqrt(Cin * 4)
    ref = F.conv2d(F.pixel_unshuffle(x.double(), 2), w1.double(), b.double())
    out = ops.conv2d(x.to(DEV), ops.pack_conv_weight(w1.to(DEV)), b.to(DEV), 1, Cout, mode=ops.CONV_UNSHUFFLE2)
    _close(out, ref, 2e-6, "unshuffle conv")


def test_conv_bad_args_raise():
    x = torch.zeros(1, 8, 8, 8, device=DEV)
    w = ops.pack_conv_weight(torch.zeros(8, 8, 3, 3, device=DEV))
    with pytest.raises(RuntimeError):
        ops.conv2d(x, w, None, 5, 8)
    with pytest.raises(RuntimeError):
        ops.conv2d(x.cpu(), w, None, 3, 8)


def test_groupnorm_via_stats_matches_torch():
    g = _g(4)
    B, C, H, W, G = 2, 64, 32, 32, 8
    x = torch.randn(B, 16, H, W, generator=g)
    w = torch.randn(C, 16, 3, 3, generator=g) / 12
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    film = torch.randn(B, 2 * C, generator=g) * 0.3
    res = torch.randn(B, C, H, W, generator=g)
    vec = torch.randn(B, C, generator=g)
    h = F.conv2d(x.double(), w.double(), None, padding=1)
    gn = F.group_norm(h, G, gamma.double(), beta.double(), 1e-5)
    sc, sh = film.double()[:, :C, None, None], film.double()[:, C:, None, None]
    ref = silu64(gn * (sc + 1) + sh) + res.double() + vec.double()[:, :, None, None]
    hd, stats = ops.conv2d(x.to(DEV), ops.pack_conv_weight(w.to(DEV)), None, 3, C, want_stats=True)
    a, b, mr = ops.gn_finalize(stats, G, H * W, gamma.to(DEV), beta.to(DEV), film=film.to(DEV), want_mean_rstd=True)
    out = ops.affine_silu_add(hd, (a, b), res=res.to(DEV), vec=vec.to(DEV))
    _close(out, ref, 5e-6, "gn+film+silu+res")
    hg = h.reshape(B, G, -1)
    _close(mr[..., 0], hg.mean(-1), 1e-5, "gn mean")
    _close(mr[..., 1], 1 / torch.sqrt(hg.var(-1, unbiased=False) + 1e-5), 1e-5, "gn rstd")


def test_linear_layernorm_time_embed():
    g = _g(5)
    R, K, N = 21, 300, 77
    x = torch.randn(R, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    res = torch.randn(R, N, generator=g)
    gs = torch.randn(N, generator=g)
    ref = res.double() + gs.double() * (F.silu(x.double()) @ w.double().T + b.double())
    ref = F.gelu(ref)
    out = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), res=res.to(DEV), gscale=gs.to(DEV), act_in=ops.ACT_SILU, act_out=ops.ACT_GELU)
    _close(out, ref, 3e-6, "linear")
    # strided slices (per-head folding uses these)
    xs = torch.randn(R, 4 * 64, generator=g)
    ws = torch.randn(256, 128, generator=g)
    outs = ops.linear(xs.to(DEV)[:, 64:128], ws.to(DEV)[:, 32:96])
    _close(outs, xs.double()[:, 64:128] @ ws.double()[:, 32:96].T, 3e-6, "linear strided")
    ga, be = torch.randn(K, generator=g), torch.randn(K, generator=g)
    ln = ops.layernorm_rows(x.to(DEV), ga.to(DEV), be.to(DEV))
    _close(ln, F.layer_norm(x.double(), (K,), ga.double(), be.double(), 1e-5), 3e-6, "layernorm")
    t = torch.tensor([0.0, 1.0, 37.0, 100.0, 998.0])
    half = 32
    freq = torch.exp(torch.arange(half, dtype=torch.float32) * (-math.log(10000.0) / (half - 1)))
    te = ops.time_embed(t.to(DEV), 64, freq.to(DEV))
    a = (t[:, None] * freq[None]).double()  # fp32 product like the oracle, then exact sin/cos
    _close(te, torch.cat([a.sin(), a.cos()], -1), 1e-6, "time_embed")
    _close(ops.time_embed(t.to(DEV), 64), torch.cat([a.sin(), a.cos()], -1), 2e-4, "time_embed (device freqs)")


def test_chan_layernorm_scoremap_gather():
    g = _g(6)
    B, C, H, W, K = 2, 64, 16, 24, 5
    x = torch.randn(B, C, H, W, generator=g) * 2 + 0.5
    ga, be = torch.randn(C, generator=g), torch.randn(C, generator=g)
    ref = F.layer_norm(x.double().permute(0, 2, 3, 1), (C,), ga.double(), be.double(), 1e-5).permute(0, 3, 1, 2)
    _close(ops.chan_layernorm(x.to(DEV), ga.to(DEV), be.to(DEV)), ref, 3e-6, "chan_layernorm")
    tv = torch.randn(B, K, C, generator=g)
    idx = torch.tensor([3, 1], dtype=torch.int32)
    ref = torch.einsum('bchw,bkc->bkhw', F.normalize(x.double(), dim=1), F.normalize(tv.double(), dim=2))
    sm, sel = ops.scoremap(x.to(DEV), tv.to(DEV), idx.to(DEV))
    _close(sm, ref, 3e-6, "scoremap")
    _close(sel, ref[torch.arange(B), idx.long()][:, None], 3e-6, "scoremap sel")
    _close(ops.gather_channel(x.to(DEV), idx.to(DEV)), x[torch.arange(B), idx.long()][:, None], 0, "gather")


@pytest.mark.parametrize("B,C,H,W,K,sliced", [(2, 64, 32, 32, 5, True), (1, 130, 8, 12, 5, False), (2, 64, 7, 9, 5, False), (3, 256, 8, 8, 8, True)])
def test_scoremap_streaming_and_scalar_forms(B, C, H, W, K, sliced):
    """rows of 16-byte-aligned length take the 4-pixels-per-thread streaming kernel (also on a channel slice of a wider buffer, as the
    UNet's skip produces it), odd sizes the one-pixel form; both against fp64, and the selected map equals the gathered channel"""
    g = _g(60 + C)
    xw = torch.randn(B, C + (16 if sliced else 0), H, W, generator=g) * 1.5 + 0.3
    tv = torch.randn(B, K, C, generator=g)
    idx = torch.randint(0, K, (B,), generator=g).to(torch.int32)
    x = xw[:, :C]
    ref = torch.einsum('bchw,bkc->bkhw', F.normalize(x.double(), dim=1), F.normalize(tv.double(), dim=2))
    sm, sel = ops.scoremap(xw.to(DEV)[:, :C], tv.to(DEV), idx.to(DEV))
    _close(sm, ref, 3e-6, "scoremap")
    assert torch.equal(sel, ops.gather_channel(sm, idx.to(DEV)))
    sm2, none = ops.scoremap(xw.to(DEV)[:, :C], tv.to(DEV))
    assert none is None and torch.equal(sm2, sm)


def _attn_ref(q, k, v, heads, scale):
    B, N, C = q.shape
    M = k.shape[1]
    q = q.reshape(B, N, heads, C // heads)
    k = k.reshape(B, M, heads, C // heads)
    v = v.reshape(B, M, heads, C // heads)
    attn = (torch.einsum('bnkc,bmkc->bknm', q, k) * scale).softmax(-1)
    return torch.einsum('bknm,bmkc->bnkc', attn, v).reshape(B, N, C)


@pytest.mark.parametrize("C,heads,H,W", [(256, 4, 32, 32), (256, 4, 8, 8), (128, 4, 12, 12), (256, 4, 20, 28)])
def test_attn_self(C, heads, H, W):
    g = _g(7)
    B, N = 2, H * W
    qkv = torch.randn(B, 3 * C, H, W, generator=g)
    qkv[0, :C] *= 3.0  # some peaky rows
    scale = (C // heads) ** -0.5
    q, k, v = qkv.double().reshape(B, 3, C, N).permute(1, 0, 3, 2)
    ref = _attn_ref(q, k, v, heads, scale).permute(0, 2, 1).reshape(B, C, H, W)
    out, lse = ops.attn_self(qkv.to(DEV), heads, scale, want_lse=True)
    _close(out, ref, 5e-6, "attn_self")
    s = torch.einsum('bnkc,bmkc->bknm', q.reshape(B, N, heads, -1), k.reshape(B, N, heads, -1)) * scale
    _close(lse, torch.logsumexp(s, -1), 5e-6, "lse")


@pytest.mark.parametrize("C,M", [(64, 1), (128, 3), (256, 7)])
def test_attn_ctx_and_tokens(C, M):
    g = _g(8)
    B, H, W, heads = 2, 16, 20, 4
    q = torch.randn(B, C, H, W, generator=g)
    k, v = torch.randn(B, M, C, generator=g), torch.randn(B, M, C, generator=g)
    scale = (C // heads) ** -0.5
    ref = _attn_ref(q.double().reshape(B, C, -1).permute(0, 2, 1), k.double(), v.double(), heads, scale)
    out = ops.attn_ctx(q.to(DEV), k.to(DEV), v.to(DEV), heads, scale)
    _close(out, ref.permute(0, 2, 1).reshape(B, C, H, W), 5e-6, "attn_ctx")
    qt = torch.randn(B, 5, C, generator=g)
    out = ops.attn_tokens(qt.to(DEV), k.to(DEV), v.to(DEV), heads, scale)
    _close(out, _attn_ref(qt.double(), k.double(), v.double(), heads, scale), 5e-6, "attn_tokens")


@pytest.mark.parametrize("N,B", [(1024, 2), (64, 1), (4096, 3), (784, 2)])
def test_smm_xattn(N, B):
    g = _g(9)
    Nq, heads, Cm = 5, 4, 256
    qf = torch.randn(B, Nq, heads, Cm, generator=g) * 0.2
    mem = torch.randn(B, Cm, N, generator=g)
    scale = 0.125
    s = torch.einsum('bqhc,bcn->bqhn', qf.double(), mem.double()) * scale
    ref = torch.einsum('bqhn,bcn->bqhc', s.softmax(-1), mem.double())
    out = ops.smm_xattn(qf.to(DEV), mem.to(DEV), scale)
    _close(out, ref, 1e-5, "smm_xattn")


# ---------------------------------------------------------------------------------------------------
def test_irsde_step_bit_exact_vs_oracle(golden_sde):
    gd = golden_sde
    sde = sde_ref.IRSDERef(0.4, T=100, sample_T=50, schedule="cosine", eps=0.01)
    mu = torch.from_numpy(gd["t64/mu"])
    x = torch.from_numpy(gd["t64/xT"])
    noises = torch.from_numpy(gd["t64/noises"])
    sde.set_mu(mu)
    g = _g(10)
    for i, t in enumerate([50, 49, 2]):
        npred = torch.randn(x.shape, generator=g)
        score = sde.get_score_from_noise(npred, t)
        kw = dict(theta=float(sde.thetas[t]), sigma=float(sde.sigmas[t]), sigma_bar=float(sde.sigma_bars[t]), dt=float(sde.dt),
                  sqrt_dt=math.sqrt(float(sde.dt)))
        for mode, ref in [(ops.SDE_STEP, sde.reverse_sde_step(x, score, t, noises[i])),
                          (ops.SDE_MEAN, sde.reverse_sde_step_mean(x, score, t)), (ops.SDE_ODE, sde.reverse_ode_step(x, score, t))]:
            out = ops.irsde_reverse_step(x.to(DEV), mu.to(DEV), npred.to(DEV), noises[i].to(DEV) if mode == ops.SDE_STEP else None,
                                         mode=mode, **kw)
            assert torch.equal(out.cpu(), ref), f"mode {mode} t {t}: max diff {(out.cpu() - ref).abs().max()}"
    # golden trajectory from the REAL reference: 3 steps with the analytic model evaluated on the host
    # (schedule scalars are taken from the golden tables: cos/exp on another host CPU may differ by an ulp)
    xs = torch.from_numpy(gd["t64/xT"]).to(DEV)
    top = gd["t64/top_steps"]
    gth, gsg, gsb = gd["cos100_s50/thetas"], gd["cos100_s50/sigmas"], gd["cos100_s50/sigma_bars"]
    gdt = float(gd["cos100_s50/dt"])
    for i, t in enumerate(gd["t64/top_ts"].tolist()):
        xc = xs.cpu()
        npred = 0.3 * xc - 0.2 * mu + (0.01 * float(t * sde.sample_scale)) * (xc * mu)
        xs = ops.irsde_reverse_step(xs, mu.to(DEV), npred.to(DEV), noises[i].to(DEV), theta=float(gth[t]), sigma=float(gsg[t]),
                                    sigma_bar=float(gsb[t]), dt=gdt, sqrt_dt=math.sqrt(gdt))
        assert np.array_equal(xs.cpu().numpy(), top[i][3])


def test_drift_step_and_mixes():
    g = _g(11)
    shp = (3, 1, 20, 24)
    x, r, e, z, cond = [torch.randn(shp, generator=g) for _ in range(5)]
    a, b, c = 0.013, 0.0071, 0.0042
    ref = sde_ref.drift_reverse_update(x, r, e, z, torch.tensor(a), torch.tensor(b), torch.tensor(c))
    out, xa = ops.drift_reverse_step(x.to(DEV), r.to(DEV), e.to(DEV), z.to(DEV), a, b, c, cond=cond.to(DEV))
    assert torch.equal(out.cpu(), ref)
    assert torch.equal(xa.cpu(), ref - cond)
    _close(ops.axpby(x.to(DEV), r.to(DEV), 1.0, -1.0), x - r, 0, "axpby")
    c0, c1, c2 = [torch.randn(3, generator=g) for _ in range(3)]
    ref = c0[:, None, None, None] * x + c1[:, None, None, None] * r + c2[:, None, None, None] * e
    _close(ops.mix3_per_sample(x.to(DEV), r.to(DEV), e.to(DEV), c0.to(DEV), c1.to(DEV), c2.to(DEV)), ref, 1e-6, "mix3")


def test_philox_bit_exact_and_randn():
    raw = ops.philox_raw(1000, DEV, seed=0x123456789ABCDEF, offset=(1 << 32) - 3).cpu().numpy().view(np.uint32)
    ref = philox_ref.philox_raw(1000, 0x123456789ABCDEF, (1 << 32) - 3)
    assert np.array_equal(raw, ref)
    z = ops.randn((1 << 16) + 3, DEV, seed=99, offset=5).cpu().numpy()
    zr = philox_ref.randn((1 << 16) + 3, 99, 5)
    assert np.abs(z - zr).max() < 1e-4
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02
    # on-device noise inside the step == injected idiff_randn noise
    shp = (2, 1, 32, 32)
    x = torch.randn(shp, device=DEV)
    mu, npred = torch.randn_like(x), torch.randn_like(x)
    kw = dict(theta=0.3, sigma=0.2, sigma_bar=0.25, dt=0.09, sqrt_dt=0.3)
    zz = ops.randn(shp, DEV, seed=7, offset=100)
    a = ops.irsde_reverse_step(x, mu, npred, zz, **kw)
    b = ops.irsde_reverse_step(x, mu, npred, None, seed=7, offset=100, **kw)
    assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------------------
# Winograd F(2x2,3x3) kernel (conv_wino.hip) vs the direct implicit-GEMM kernel and the fp64 reference
def _pack(w, wino, transpose=False, wino4=False):
    old, old4 = ops.WINOGRAD, ops.WINOGRAD4
    ops.WINOGRAD, ops.WINOGRAD4 = wino, wino4
    try:
        p = ops.pack_conv_weight(w, transpose=transpose)
    finally:
        ops.WINOGRAD, ops.WINOGRAD4 = old, old4
    assert hasattr(p, "wino") == wino and hasattr(p, "wino4") == wino4
    return p


@pytest.mark.parametrize("B,C0,C1,Cout,H,W,variant", [
    (2, 64, 0, 64, 32, 32, "plain"),
    (1, 8, 0, 128, 8, 64, "plain"),         # one chunk, two channel blocks, 1x2 patches
    (2, 64, 0, 64, 16, 96, "prologue"),
    (1, 24, 40, 192, 24, 32, "concat"),
    (2, 32, 0, 64, 16, 16, "upsample"),     # out 32x32
    (1, 128, 0, 256, 64, 64, "epilogue"),
    (2, 32, 0, 80, 8, 32, "plain"),         # Cout % 64 = 16: partial last channel block
    (1, 64, 0, 144, 16, 32, "epilogue"),    # partial block with residual / aux / stats
    # image not a multiple of the 8x32 patch: partial patches at the right / bottom border (the reference's native 224 = 7 x 32 and
    # its lower levels 112 / 56), masked stores and statistics
    (2, 32, 0, 64, 28, 56, "plain"),
    (1, 64, 0, 64, 14, 112, "prologue"),
    (2, 16, 0, 48, 6, 36, "epilogue"),
    (1, 16, 0, 32, 7, 28, "upsample"),      # out 14x56
    (1, 24, 16, 64, 10, 40, "concat"),
    (2, 32, 0, 64, 28, 28, "epilogue"),     # narrower than a patch (the 224 / 8 level)
    (1, 16, 0, 32, 12, 24, "plain"),
    # several items per workgroup, one and two chunks per item: the software pipeline runs across item boundaries with up to
    # three items in flight (four table parities)
    (11, 8, 0, 64, 64, 64, "epilogue"),
    (10, 16, 0, 128, 64, 64, "prologue"),
    (9, 40, 0, 64, 64, 96, "plain"),
])
def test_conv_winograd_matches_direct_and_fp64(B, C0, C1, Cout, H, W, variant):
    g = _g(11)
    Cin = C0 + C1
    x0 = torch.randn(B, C0, H, W, generator=g)
    x1 = torch.randn(B, C1, H, W, generator=g) if C1 else None
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g)
    kw = {}
    xin = x0.double() if x1 is None else torch.cat([x0, x1], 1).double()
    if variant == "prologue":
        pa, pb = torch.randn(B, C0, generator=g), torch.randn(B, C0, generator=g)
        xin = silu64(pa.double()[:, :, None, None] * xin + pb.double()[:, :, None, None])
        kw["pro"] = (pa.to(DEV), pb.to(DEV))
    if variant == "upsample":
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
        kw["mode"] = ops.CONV_UPSAMPLE2
    if x1 is not None:
        kw["src1"] = x1.to(DEV)
    raw = F.conv2d(xin, w.double(), b.double(), padding=1)
    ref = raw
    Ho, Wo = raw.shape[2:]
    if variant == "epilogue":
        res = torch.randn(B, Cout, Ho, Wo, generator=g)
        vec = torch.randn(B, Cout, generator=g)
        aux = torch.randn(B, Cout, Ho, Wo, generator=g)
        aa, ab = torch.randn(B, Cout, generator=g), torch.randn(B, Cout, generator=g)
        ref = raw + res.double() + vec.double()[:, :, None, None] + silu64(aa.double()[:, :, None, None] * aux.double() + ab.double()[:, :, None, None])
        kw.update(res=res.to(DEV), vec=vec.to(DEV), aux=(aux.to(DEV), aa.to(DEV), ab.to(DEV)))
    wd = w.to(DEV)
    lib = ops._lib.load()
    out_w, st_w = ops.conv2d(x0.to(DEV), _pack(wd, True), b.to(DEV), 3, Cout, want_stats=True, **kw)
    assert lib.idiff_conv2d_last_algo() == 1, "the Winograd kernel did not run"
    out_d, st_d = ops.conv2d(x0.to(DEV), _pack(wd, False), b.to(DEV), 3, Cout, want_stats=True, **kw)
    assert lib.idiff_conv2d_last_algo() == 0
    _close(out_d, ref, 2e-6, "direct")
    _close(out_w, ref, 6e-6, "winograd")          # a few ulps of reassociation more than the direct kernel
    assert st_w.shape == st_d.shape
    _close(st_w.sum(1)[..., 0], raw.sum(dim=(2, 3)), 1e-5, "winograd stats sum")
    _close(st_w.sum(1)[..., 1], (raw ** 2).sum(dim=(2, 3)), 1e-5, "winograd stats sumsq")
    _close(st_w, st_d.cpu(), 2e-5, "per-patch stats layout")


def test_conv_winograd_data_gradient_pack():
    """The transposed pack feeds the data-gradient conv: dX = conv(dY, flip(W)^T)."""
    g = _g(12)
    B, Cin, Cout, H, W = 2, 64, 128, 16, 32
    dy = torch.randn(B, Cout, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    ref = F.conv_transpose2d(dy.double(), w.double(), padding=1)
    wd = w.to(DEV)
    dx_w = ops.conv2d(dy.to(DEV), _pack(wd, True, transpose=True), None, 3, Cin)
    dx_d = ops.conv2d(dy.to(DEV), _pack(wd, False, transpose=True), None, 3, Cin)
    _close(dx_d, ref, 2e-6, "direct dgrad")
    _close(dx_w, ref, 6e-6, "winograd dgrad")
    lib = ops._lib.load()
    dx_4 = ops.conv2d(dy.to(DEV), _pack(wd, True, transpose=True, wino4=True), None, 3, Cin, algo=ops.CONV_ALGO_WINOGRAD4)
    assert lib.idiff_conv2d_last_algo() == 3
    _close(dx_4, ref, 4e-5, "winograd4 dgrad")


def test_conv_winograd_random_shapes_match_direct():
    """Seeded sweep over the Winograd kernel's eligibility lattice (channel blocks, patch grid, gather variants, epilogue
    terms): the two kernels behind idiff_conv2d_fwd must agree everywhere."""
    rng = np.random.RandomState(7)
    lib = ops._lib.load()
    for case in range(24):
        B = int(rng.randint(1, 4))
        C0 = 8 * int(rng.randint(1, 13))
        two = bool(rng.randint(0, 2)) and case % 3 == 1
        C1 = 8 * int(rng.randint(1, 7)) if two else 0
        Cout = 16 * int(rng.randint(1, 13))
        H = 8 * int(rng.randint(1, 4))
        W = 32 * int(rng.randint(1, 4))
        pro = (not two) and case % 3 == 2
        g = _g(100 + case)
        x0 = torch.randn(B, C0, H, W, generator=g).to(DEV)
        x1 = torch.randn(B, C1, H, W, generator=g).to(DEV) if two else None
        w = (torch.randn(Cout, C0 + C1, 3, 3, generator=g) / math.sqrt(9 * (C0 + C1))).to(DEV)
        bias = torch.randn(Cout, generator=g).to(DEV)
        kw = {}
        if pro:
            kw["pro"] = (torch.randn(B, C0, generator=g).to(DEV), torch.randn(B, C0, generator=g).to(DEV))
        if case % 2:
            kw["res"] = torch.randn(B, Cout, H, W, generator=g).to(DEV)
            kw["vec"] = torch.randn(B, Cout, generator=g).to(DEV)
        if case % 4 == 0:
            kw["aux"] = (torch.randn(B, Cout, H, W, generator=g).to(DEV), torch.randn(B, Cout, generator=g).to(DEV), torch.randn(B, Cout, generator=g).to(DEV))
        ow, sw = ops.conv2d(x0, _pack(w, True), bias, 3, Cout, src1=x1, want_stats=True, **kw)
        assert lib.idiff_conv2d_last_algo() == 1, (case, B, C0, C1, Cout, H, W)
        od, sd = ops.conv2d(x0, _pack(w, False), bias, 3, Cout, src1=x1, want_stats=True, **kw)
        _close(ow, od.cpu(), 8e-6, f"case {case}: B={B} C0={C0} C1={C1} Cout={Cout} H={H} W={W} pro={pro}")
        _close(sw, sd.cpu(), 3e-5, f"case {case}: stats")


# ---------------------------------------------------------------------------------------------------
# Winograd F(4x4,3x3) kernel (conv_wino4.hip): 16x32-pixel items, 6x6 transforms.  fp32 with transform constants up to 8:
# about one decimal digit less than F(2x2,3x3) (tolerance 4e-5 of the output range, measured ~1e-5).
@pytest.fixture(params=[ops.CONV_ALGO_WINOGRAD4, ops.CONV_ALGO_WINOGRAD4H], ids=["patch16x32", "halfpatch8x32"])
def force_wino4(request):
    """every 3x3 conv of the test asks for one of the two F(4x4,3x3) kernels (conv_wino4.hip: 16x32-pixel items, one workgroup per
    CU; conv_wino4h.hip: 8x32-pixel items, two workgroups per CU, weights straight into the A operand) through the per-call
    idiff_conv_desc.algo_request (the library's own choice keeps small layers on other kernels)"""
    with ops.request_conv3x3_algo(request.param):
        lib = ops._lib.load()
        lib.expected_algo = request.param
        yield lib


@pytest.mark.parametrize("B,C0,C1,Cout,H,W,variant", [
    (2, 64, 0, 64, 32, 32, "plain"),
    (1, 16, 0, 128, 16, 64, "plain"),       # four chunks (the fewest the 16x32 kernel takes), two channel blocks, 1x2 patches
    (2, 64, 0, 64, 16, 96, "prologue"),
    (1, 24, 40, 192, 32, 32, "concat"),
    (2, 32, 0, 64, 16, 16, "upsample"),     # out 32x32
    (1, 128, 0, 256, 64, 64, "epilogue"),
    (2, 32, 0, 80, 16, 32, "plain"),        # Cout % 64 = 16: partial last channel block
    (1, 64, 0, 144, 16, 32, "epilogue"),    # partial block with residual / aux / stats
    # image not a multiple of the 16x32 patch: partial patches at the right / bottom border, masked stores and statistics
    (2, 32, 0, 64, 28, 56, "plain"),
    (1, 64, 0, 64, 8, 112, "prologue"),     # half a patch high: the lower half-patch is entirely outside
    (2, 16, 0, 48, 12, 36, "epilogue"),
    (1, 16, 0, 32, 14, 28, "upsample"),     # out 28x56
    (1, 24, 16, 64, 20, 40, "concat"),
    (2, 32, 0, 64, 28, 28, "epilogue"),     # narrower than a patch (the 224 / 8 level)
    (1, 16, 0, 32, 24, 24, "plain"),
    (1, 16, 0, 64, 56, 56, "prologue"),     # 56 = 3.5 patches high
    # several items per workgroup, four to ten chunks per item: the software pipeline runs across item boundaries
    (21, 16, 0, 64, 64, 64, "epilogue"),
    (20, 16, 0, 128, 64, 64, "prologue"),
    (17, 40, 0, 64, 64, 96, "plain"),
    # partial patches AND more items than workgroups: the stream crosses item boundaries between whole and partial patches
    (40, 16, 0, 64, 56, 56, "plain"),
    (36, 8, 16, 64, 40, 56, "concat"),
    (34, 16, 0, 64, 56, 40, "prologue"),
])
def test_conv_winograd4_matches_direct_and_fp64(force_wino4, B, C0, C1, Cout, H, W, variant):
    _check_wino4_case(force_wino4, B, C0, C1, Cout, H, W, variant)


def _check_wino4_case(lib, B, C0, C1, Cout, H, W, variant):
    g = _g(21)
    Cin = C0 + C1
    x0 = torch.randn(B, C0, H, W, generator=g)
    x1 = torch.randn(B, C1, H, W, generator=g) if C1 else None
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g)
    kw = {}
    xin = x0.double() if x1 is None else torch.cat([x0, x1], 1).double()
    if variant == "prologue":
        pa, pb = torch.randn(B, C0, generator=g), torch.randn(B, C0, generator=g)
        xin = silu64(pa.double()[:, :, None, None] * xin + pb.double()[:, :, None, None])
        kw["pro"] = (pa.to(DEV), pb.to(DEV))
    if variant == "upsample":
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
        kw["mode"] = ops.CONV_UPSAMPLE2
    if x1 is not None:
        kw["src1"] = x1.to(DEV)
    raw = F.conv2d(xin, w.double(), b.double(), padding=1)
    ref = raw
    Ho, Wo = raw.shape[2:]
    if variant == "epilogue":
        res = torch.randn(B, Cout, Ho, Wo, generator=g)
        vec = torch.randn(B, Cout, generator=g)
        aux = torch.randn(B, Cout, Ho, Wo, generator=g)
        aa, ab = torch.randn(B, Cout, generator=g), torch.randn(B, Cout, generator=g)
        ref = raw + res.double() + vec.double()[:, :, None, None] + silu64(aa.double()[:, :, None, None] * aux.double() + ab.double()[:, :, None, None])
        kw.update(res=res.to(DEV), vec=vec.to(DEV), aux=(aux.to(DEV), aa.to(DEV), ab.to(DEV)))
    wd = w.to(DEV)

    def pk():
        return _pack(wd, True, wino4=True)
    out_w, st_w = ops.conv2d(x0.to(DEV), pk(), b.to(DEV), 3, Cout, want_stats=True, **kw)
    assert lib.idiff_conv2d_last_algo() == lib.expected_algo, "the requested F(4x4,3x3) kernel did not run"
    out_d, st_d = ops.conv2d(x0.to(DEV), _pack(wd, False), b.to(DEV), 3, Cout, want_stats=True, algo=ops.CONV_ALGO_DIRECT, **kw)
    assert lib.idiff_conv2d_last_algo() == 0
    _close(out_w, ref, 4e-5, "winograd4")
    assert st_w.shape == st_d.shape
    _close(st_w.sum(1)[..., 0], raw.sum(dim=(2, 3)), 4e-5, "winograd4 stats sum")
    _close(st_w.sum(1)[..., 1], (raw ** 2).sum(dim=(2, 3)), 4e-5, "winograd4 stats sumsq")
    _close(st_w, st_d.cpu(), 1e-4, "per-patch stats layout")
    # without statistics, and the launch is repeatable bit for bit
    out_2 = ops.conv2d(x0.to(DEV), pk(), b.to(DEV), 3, Cout, **kw)
    assert torch.equal(out_2, out_w)


def test_conv_winograd4_policy_and_fallback():
    """Small levels (fewer than 16 items per sample, whatever the batch) stay on F(2x2,3x3); shapes outside the 4x4 tiling
    fall back."""
    lib = ops._lib.load()
    g = _g(22)
    w = (torch.randn(64, 32, 3, 3, generator=g) / 17.0).to(DEV)
    wpk = _pack(w, True, wino4=True)
    x = torch.randn(2, 32, 32, 32, generator=g).to(DEV)
    ops.conv2d(x, wpk, None, 3, 64)
    assert lib.idiff_conv2d_last_algo() == 1           # 2 items per sample
    ops.conv2d(x, wpk, None, 3, 64, algo=ops.CONV_ALGO_WINOGRAD4)   # asked for by name: the threshold is waived
    assert lib.idiff_conv2d_last_algo() == 3
    x2 = torch.randn(1, 32, 30, 36, generator=g).to(DEV)   # H % 4 != 0
    ref = F.conv2d(x2.double().cpu(), w.double().cpu(), padding=1)
    with pytest.raises(ops._lib.IdiffError):               # a hard request for a shape the kernel does not tile fails loudly
        ops.conv2d(x2, wpk, None, 3, 64, algo=ops.CONV_ALGO_WINOGRAD4)
    with ops.request_conv3x3_algo(ops.CONV_ALGO_WINOGRAD4):  # the scope form is a preference: the library's choice elsewhere
        _close(ops.conv2d(x2, wpk, None, 3, 64), ref, 6e-6, "fallback")
    assert lib.idiff_conv2d_last_algo() == 1
    o_d = ops.conv2d(x, wpk, None, 3, 64, algo=ops.CONV_ALGO_DIRECT)
    assert lib.idiff_conv2d_last_algo() == 0
    _close(o_d, F.conv2d(x.double().cpu(), w.double().cpu(), padding=1), 2e-6, "direct by request")
    big = torch.randn(16, 32, 64, 64, generator=g).to(DEV)   # 8 items of 16x32 per sample (the batch does not count), 16 of 8x32:
    ref_big = F.conv2d(big.double().cpu(), w.double().cpu(), padding=1)
    _close(ops.conv2d(big, wpk, None, 3, 64), ref_big, 4e-5, "half-patch kernel by the library's choice")
    assert lib.idiff_conv2d_last_algo() == 4               # the half-patch form of the F(4x4,3x3) kernel
    two = ops.conv2d(big[:, :16].contiguous(), wpk, None, 3, 64, src1=big[:, 16:].contiguous())   # same conv as a virtual concat
    assert lib.idiff_conv2d_last_algo() == 4
    _close(two, ref_big, 4e-5, "half-patch kernel, two sources")
    big3 = torch.randn(2, 32, 128, 128, generator=g).to(DEV)
    ops.conv2d(big3[:, :16].contiguous(), wpk, None, 3, 64, src1=big3[:, 16:].contiguous())
    assert lib.idiff_conv2d_last_algo() == 3               # r04: from 16 items per sample up the 16x32 kernel, two sources or one
    big = torch.randn(3, 32, 128, 128, generator=g).to(DEV)   # 32 items per sample
    out = ops.conv2d(big, wpk, None, 3, 64)
    assert lib.idiff_conv2d_last_algo() == 3
    _close(out, F.conv2d(big.double().cpu(), w.double().cpu(), padding=1), 4e-5, "winograd4 128x128")


def test_conv_winograd4_random_shapes_match_direct(force_wino4):
    """Seeded sweep over the F(4x4,3x3) kernel's eligibility lattice (channel blocks incl. partial ones, whole and partial 16x32
    patches, one to many items per workgroup, gather variants, epilogue terms): it must agree with the direct kernel everywhere."""
    lib = force_wino4
    rng = np.random.RandomState(17)
    for case in range(28):
        B = int(rng.randint(1, 6))
        C0 = 8 * int(rng.randint(1, 13))
        two = bool(rng.randint(0, 2)) and case % 3 == 1
        C1 = 4 * int(rng.randint(1, 9)) if two else 0
        if (C0 + C1) % 8:
            C1 += 4
        if C0 + C1 < 16:   # the 16x32-item kernel streams four chunks ahead: Cin >= 16
            C0 += 8
        Cout = 16 * int(rng.randint(1, 13))
        H = 4 * int(rng.randint(1, 13))
        W = 4 * int(rng.randint(6, 25))
        up = (not two) and case % 7 == 3
        pro = (not two) and (not up) and case % 3 == 2
        g = _g(300 + case)
        x0 = torch.randn(B, C0, H, W, generator=g).to(DEV)
        x1 = torch.randn(B, C1, H, W, generator=g).to(DEV) if two else None
        w = (torch.randn(Cout, C0 + C1, 3, 3, generator=g) / math.sqrt(9 * (C0 + C1))).to(DEV)
        bias = torch.randn(Cout, generator=g).to(DEV)
        Ho, Wo = (2 * H, 2 * W) if up else (H, W)
        kw = {}
        if up:
            kw["mode"] = ops.CONV_UPSAMPLE2
        if pro:
            kw["pro"] = (torch.randn(B, C0, generator=g).to(DEV), torch.randn(B, C0, generator=g).to(DEV))
        if case % 2:
            kw["res"] = torch.randn(B, Cout, Ho, Wo, generator=g).to(DEV)
            kw["vec"] = torch.randn(B, Cout, generator=g).to(DEV)
        if case % 4 == 0:
            kw["aux"] = (torch.randn(B, Cout, Ho, Wo, generator=g).to(DEV), torch.randn(B, Cout, generator=g).to(DEV), torch.randn(B, Cout, generator=g).to(DEV))
        tag = f"case {case}: B={B} C0={C0} C1={C1} Cout={Cout} H={H} W={W} up={up} pro={pro}"
        ow, sw = ops.conv2d(x0, _pack(w, True, wino4=True), bias, 3, Cout, src1=x1, want_stats=True, **kw)
        assert lib.idiff_conv2d_last_algo() == lib.expected_algo, tag
        od, sd = ops.conv2d(x0, _pack(w, False), bias, 3, Cout, src1=x1, want_stats=True, **kw)
        assert lib.idiff_conv2d_last_algo() == 0, tag
        _close(ow, od.cpu(), 4e-5, tag)
        _close(sw, sd.cpu(), 1e-4, tag + " stats")


def test_conv_winograd4_bits_do_not_depend_on_the_batch(force_wino4):
    """A sample's result is the same bit pattern whether it is convolved alone or inside a batch (persistent items of several
    samples per workgroup, item order, kernel choice): the property the B=16 / B=5 chain tests rest on, at op level."""
    lib = force_wino4
    g = _g(23)
    B, C0, Cout, H, W = 5, 40, 80, 28, 56
    x = torch.randn(B, C0, H, W, generator=g).to(DEV)
    w = (torch.randn(Cout, C0, 3, 3, generator=g) / math.sqrt(9 * C0)).to(DEV)
    bias = torch.randn(Cout, generator=g).to(DEV)
    pro = (torch.randn(B, C0, generator=g).to(DEV), torch.randn(B, C0, generator=g).to(DEV))
    wp = _pack(w, True, wino4=True)
    full, st = ops.conv2d(x, wp, bias, 3, Cout, pro=pro, want_stats=True)
    assert lib.idiff_conv2d_last_algo() == lib.expected_algo
    for b in (0, 3, 4):
        one, s1 = ops.conv2d(x[b:b + 1].contiguous(), wp, bias, 3, Cout, pro=(pro[0][b:b + 1].contiguous(), pro[1][b:b + 1].contiguous()), want_stats=True)
        assert lib.idiff_conv2d_last_algo() == lib.expected_algo
        assert torch.equal(one[0], full[b]) and torch.equal(s1[0], st[b]), b


@pytest.mark.parametrize("B,C0,C1,Cout,H,W,G,film,pro,algo", [
    (16, 64, 0, 64, 64, 64, 8, True, False, None),    # half-patch kernel by the library's choice
    (16, 64, 0, 64, 64, 64, 8, True, False, 3),       # the same on the 16x32-item kernel
    (3, 64, 0, 64, 128, 128, 8, False, True, None),   # 16x32-item kernel, prologue instantiation
    (2, 32, 48, 128, 32, 64, 8, True, False, None),   # two sources: half-patch kernel, 2 workgroups per CU
    (5, 128, 0, 256, 32, 32, 8, True, False, None),   # half-patch kernel below 16 items of 16x32 per sample
    (1, 16, 0, 32, 28, 28, 8, False, False, 4),       # 4 workgroups, 8 (sample, group) pairs
    (1, 16, 0, 32, 28, 28, 4, False, False, None),    # F(2x2,3x3): the library enqueues the separate finalize launch
    (2, 16, 0, 32, 16, 16, 8, True, False, None),     # direct kernel: likewise
])
def test_conv_groupnorm_finalize_riding_on_the_conv_call_equals_separate_launch(B, C0, C1, Cout, H, W, G, film, pro, algo):
    """idiff_conv_desc.gn_*: the GroupNorm(+FiLM) finalize riding on the conv call (one C call: the library enqueues the finalize
    launch behind whichever conv kernel it picked) gives the SAME BITS as idiff_gn_finalize on the conv's statistics, launch after
    launch.  (r03 / r04 also ran the finalize as the tail of the F(4x4,3x3) launches; removed as slower -- DESIGN.md section 8.)"""
    lib = ops._lib.load()
    g = _g(500 + B + Cout)
    x0 = torch.randn(B, C0, H, W, generator=g).to(DEV)
    x1 = torch.randn(B, C1, H, W, generator=g).to(DEV) if C1 else None
    w = (torch.randn(Cout, C0 + C1, 3, 3, generator=g) / math.sqrt(9 * (C0 + C1))).to(DEV)
    bias = torch.randn(Cout, generator=g).to(DEV)
    gamma, beta = torch.randn(Cout, generator=g).to(DEV), torch.randn(Cout, generator=g).to(DEV)
    fl = (torch.randn(B, 2 * Cout + 8, generator=g) * 0.3).to(DEV)[:, :2 * Cout] if film else None   # row-strided, as the UNet's films are
    kw = dict(src1=x1, algo=algo)
    if pro:
        kw["pro"] = (torch.rand(B, C0, generator=g).to(DEV) + 0.5, torch.randn(B, C0, generator=g).to(DEV) * 0.1)
    wp = ops.pack_conv_weight(w)
    out_ref, st = ops.conv2d(x0, wp, bias, 3, Cout, want_stats=True, **kw)
    algo = lib.idiff_conv2d_last_algo()
    a_ref, b_ref, mr_ref = ops.gn_finalize(st, G, H * W, gamma, beta, film=fl, eps=1e-5, want_mean_rstd=True)
    for rep in range(3):
        out, (a, b, mr) = ops.conv2d(x0, wp, bias, 3, Cout, gn=dict(groups=G, gamma=gamma, beta=beta, film=fl, eps=1e-5, want_mean_rstd=True), **kw)
        assert lib.idiff_conv2d_last_algo() == algo
        assert torch.equal(out, out_ref)
        assert torch.equal(a, a_ref) and torch.equal(b, b_ref) and torch.equal(mr, mr_ref), (rep, algo)
    out, (a, b) = ops.conv2d(x0, wp, bias, 3, Cout, gn=dict(groups=G, gamma=gamma, beta=beta, film=fl, eps=1e-5), **kw)
    assert torch.equal(a, a_ref) and torch.equal(b, b_ref)
    print(f"conv algo {algo}: finalize on the conv call == separate launch (bitwise), B={B} Cout={Cout} {H}x{W}")


# ---- 1x1 conv on the bf16 matrix cores, fp32 operands split three ways (csrc/conv1x1_x3.hip) ------------------------------------------
@pytest.mark.parametrize("B,C0,C1,Cout,H,W,variant", [
    (2, 64, 64, 64, 32, 32, "res"),        # the level-0 residual 1x1 (virtual concat), 4 chunks of 32
    (1, 64, 80, 64, 64, 64, "plain"),      # Cin = 144: last chunk half empty; the sources meet inside a chunk (octet boundary)
    (2, 128, 80, 128, 16, 32, "vec_aux"),  # 208 -> 128, two channel blocks, per-(b,c) vector and the "+ silu(a*aux+b)" term
    (1, 256, 0, 768, 32, 32, "plain"),     # single source (the mid-attention qkv projection), 12 channel blocks
    (3, 32, 0, 64, 8, 32, "res"),          # one chunk, 256 pixels per sample: one tile
])
def test_conv1x1_bf16x3_matches_fp64_and_the_f32_kernel(B, C0, C1, Cout, H, W, variant):
    """IDIFF_CONV_ALGO_X3: six bf16 MFMAs per product on operands that sum to the fp32 value exactly -- held to the f32 kernel's own
    tolerance against fp64 (2e-6 of the output range), and compared with the f32-matrix-core kernel on the same call."""
    g = _g(11)
    Cin = C0 + C1
    x0 = torch.randn(B, C0, H, W, generator=g) * 3.0
    x1 = torch.randn(B, C1, H, W, generator=g) * 0.5 if C1 else None
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / math.sqrt(Cin)
    bias = torch.randn(Cout, generator=g)
    xin = torch.cat([x0, x1], 1) if C1 else x0
    ref = F.conv2d(xin.double(), w.double(), bias.double())
    kw = {}
    if variant == "res":
        r = torch.randn(B, Cout, H, W, generator=g)
        ref = ref + r.double()
        kw["res"] = r.to(DEV)
    if variant == "vec_aux":
        vec = torch.randn(B, Cout, generator=g)
        aux, aa, ab = torch.randn(B, Cout, H, W, generator=g), torch.randn(B, Cout, generator=g), torch.randn(B, Cout, generator=g)
        ref = ref + vec.double()[:, :, None, None] + silu64(aa.double()[:, :, None, None] * aux.double() + ab.double()[:, :, None, None])
        kw.update(vec=vec.to(DEV), aux=(aux.to(DEV), aa.to(DEV), ab.to(DEV)))
    wpk = ops.pack_conv_weight(w.to(DEV))
    assert getattr(wpk, "x3", None) is not None
    args = (x0.to(DEV), wpk, bias.to(DEV), 1, Cout)
    src1 = x1.to(DEV) if C1 else None
    out = ops.conv2d(*args, src1=src1, **kw)
    assert ops._lib.load().idiff_conv2d_last_algo() == ops.CONV_ALGO_X3
    _close(out, ref, 2e-6, "conv1x1 bf16x3 vs fp64")
    out32 = ops.conv2d(*args, src1=src1, algo=ops.CONV_ALGO_DIRECT, **kw)
    assert ops._lib.load().idiff_conv2d_last_algo() == ops.CONV_ALGO_DIRECT
    _close(out, out32, 2e-6, "conv1x1 bf16x3 vs the f32 kernel")
    e3 = float((out.cpu().double() - ref).abs().max())
    e32 = float((out32.cpu().double() - ref).abs().max())
    print(f"1x1 {Cin}->{Cout} {H}x{W}: max err vs fp64  bf16x3 {e3:.3e}   f32 MFMA {e32:.3e}")
    assert e3 < 3 * e32 + 1e-7
    again = ops.conv2d(*args, src1=src1, **kw)
    assert torch.equal(out, again)


def test_conv1x1_bf16x3_extreme_magnitudes_and_policy():
    """operands spanning 2^-60 .. 2^60 (bf16 keeps fp32's exponent range: no overflow, unlike an fp16 split); layers the kernel does not
    cover stay on the f32 kernel, and asking for it by name there is an error"""
    g = _g(12)
    B, Cin, Cout, H, W = 1, 64, 64, 8, 32
    x = torch.randn(B, Cin, H, W, generator=g) * torch.exp2(torch.randint(-60, 60, (B, Cin, 1, 1), generator=g).float())
    w = torch.randn(Cout, Cin, 1, 1, generator=g) * torch.exp2(-torch.randint(-60, 60, (1, Cin, 1, 1), generator=g).float())
    ref = F.conv2d(x.double(), w.double())
    wpk = ops.pack_conv_weight(w.to(DEV))
    out = ops.conv2d(x.to(DEV), wpk, None, 1, Cout)
    assert ops._lib.load().idiff_conv2d_last_algo() == ops.CONV_ALGO_X3 and torch.isfinite(out).all()
    # products differ by 2^120 here: compare with the f32 kernel's error rather than with a fixed fraction of the range
    out32 = ops.conv2d(x.to(DEV), wpk, None, 1, Cout, algo=ops.CONV_ALGO_DIRECT)
    e3, e32 = float((out.cpu().double() - ref).abs().max()), float((out32.cpu().double() - ref).abs().max())
    assert e3 <= 4 * e32 + 1e-6 * float(ref.abs().max()), (e3, e32)
    # 40 input pixels do not tile by 256: the f32 kernel serves it; a hard request is refused
    xs = torch.randn(1, 64, 5, 8, generator=g)
    ops.conv2d(xs.to(DEV), wpk, None, 1, Cout)
    assert ops._lib.load().idiff_conv2d_last_algo() == ops.CONV_ALGO_DIRECT
    with pytest.raises(Exception, match="bf16x3"):
        ops.conv2d(xs.to(DEV), wpk, None, 1, Cout, algo=ops.CONV_ALGO_X3)


@pytest.mark.parametrize("B,C0,Cout,H,W,variant", [
    (2, 64, 64, 32, 64, "plain"),    # the level-0 downsample (256 virtual channels -> 64), 2 output rows of 32 per... tile = 8 rows
    (1, 128, 256, 64, 64, "res"),    # 512 -> 256, four channel blocks
    (3, 16, 64, 16, 128, "plain"),   # 64 virtual channels: two chunks; wide rows (a 256-pixel tile = four output rows of 64)
])
def test_conv1x1_bf16x3_pixel_unshuffle(B, C0, Cout, H, W, variant):
    """IDIFF_CONV_UNSHUFFLE2 on the bf16x3 kernel: the gather re-indexes eight 16-byte row pieces in registers"""
    g = _g(13)
    x = torch.randn(B, C0, H, W, generator=g) * 2.0
    w = torch.randn(Cout, 4 * C0, 1, 1, generator=g) / math.sqrt(4 * C0)
    bias = torch.randn(Cout, generator=g)
    ref = F.conv2d(F.pixel_unshuffle(x.double(), 2), w.double(), bias.double())
    kw = {}
    if variant == "res":
        r = torch.randn(B, Cout, H // 2, W // 2, generator=g)
        ref = ref + r.double()
        kw["res"] = r.to(DEV)
    wpk = ops.pack_conv_weight(w.to(DEV))
    out = ops.conv2d(x.to(DEV), wpk, bias.to(DEV), 1, Cout, mode=ops.CONV_UNSHUFFLE2, **kw)
    assert ops._lib.load().idiff_conv2d_last_algo() == ops.CONV_ALGO_X3
    _close(out, ref, 2e-6, "unshuffle conv1x1 bf16x3 vs fp64")
    out32 = ops.conv2d(x.to(DEV), wpk, bias.to(DEV), 1, Cout, mode=ops.CONV_UNSHUFFLE2, algo=ops.CONV_ALGO_DIRECT, **kw)
    assert ops._lib.load().idiff_conv2d_last_algo() == ops.CONV_ALGO_DIRECT
    _close(out, out32, 2e-6, "unshuffle conv1x1 bf16x3 vs the f32 kernel")


@pytest.mark.parametrize("Cin,Cout,H,W,ks,mode,transpose", [
    (64, 64, 64, 64, 3, 0, False),    # F(4x4,3x3), 16x32 items
    (64, 128, 16, 32, 3, 0, True),    # half-patch kernel, data-gradient pack
    (32, 64, 6, 24, 3, 0, False),     # F(2x2,3x3)
    (64, 5, 32, 32, 3, 0, False),     # direct
    (128, 64, 32, 32, 1, 0, False),   # 1x1 on the bf16 matrix cores
    (64, 128, 32, 32, 1, 0, True),    # its data-gradient pack (transposed matrix)
    (16, 64, 32, 32, 1, 2, False),    # pixel-unshuffle 1x1
    (24, 40, 16, 16, 1, 0, False),    # 1x1, direct
    (32, 64, 8, 16, 3, 1, False),     # upsample
])
def test_conv_weight_packed_at_the_call_fills_the_image_the_launched_kernel_reads(Cin, Cout, H, W, ks, mode, transpose):
    """ops.LazyConvWeight (the training path's weights): idiff_conv2d_plan names the kernel, only its image is packed, the result is
    bit-identical to the call with every image packed, and the kernel that ran is the planned one (conv2d raises otherwise)."""
    g = torch.Generator().manual_seed(77)
    B = 2
    C0 = Cin // 4 if mode == ops.CONV_UNSHUFFLE2 else Cin
    x = torch.randn(B, C0, H, W, generator=g).to(DEV)
    w = (torch.randn((Cin, Cout, ks, ks) if transpose else (Cout, Cin, ks, ks), generator=g) / math.sqrt(Cin * ks * ks)).to(DEV)
    full = ops.conv2d(x, ops.pack_conv_weight(w, transpose=transpose), None, ks, Cout, mode=mode)
    a_full = ops._lib.load().idiff_conv2d_last_algo()
    lazy = ops.conv2d(x, ops.LazyConvWeight(w, transpose=transpose), None, ks, Cout, mode=mode)
    assert ops._lib.load().idiff_conv2d_last_algo() == a_full
    assert torch.equal(full, lazy)
