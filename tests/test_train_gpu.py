"""Training-path parity on the GPU: every autograd Function's backward (HIP kernels) against torch autograd of the
plain-PyTorch formula on the CPU, then whole-UNet parameter gradients and two Adam steps against the oracle."""
import math

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from instancediff_amd import ops, pipeline, train_ops as T  # noqa: E402
from instancediff_amd.utils.synthetic import make_batch  # noqa: E402
from oracle import sde_ref, unet_ref  # noqa: E402

DEV = "cuda"


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def _leaf(t):
    return t.clone().to(DEV).requires_grad_(True)


@pytest.mark.parametrize("C0,C1,Cout,H,W,ks,mode", [
    (24, 0, 64, 16, 16, 3, 0), (16, 24, 40, 32, 32, 3, 0), (32, 0, 64, 8, 8, 3, 1), (16, 0, 64, 16, 32, 1, 2),
    (40, 8, 64, 16, 16, 1, 0), (2, 0, 64, 32, 32, 7, 0), (64, 0, 5, 32, 32, 3, 0), (5, 0, 16, 16, 16, 3, 0)])
def test_conv_fn_backward(C0, C1, Cout, H, W, ks, mode):
    g = _g(1)
    B = 2
    x0 = torch.randn(B, C0, H, W, generator=g)
    x1 = torch.randn(B, C1, H, W, generator=g) if C1 else None
    cin = (C0 * 4 if mode == 2 else C0) + C1
    w = torch.randn(Cout, cin, ks, ks, generator=g) / math.sqrt(cin * ks * ks)
    b = torch.randn(Cout, generator=g)
    # reference
    r0, r1, rw, rb = x0.double().requires_grad_(True), (x1.double().requires_grad_(True) if C1 else None), w.double().requires_grad_(True), \
        b.double().requires_grad_(True)
    xin = torch.cat([r0, r1], 1) if C1 else r0
    if mode == 1:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    elif mode == 2:
        xin = F.pixel_unshuffle(xin, 2)
    ref = F.conv2d(xin, rw, rb, padding=ks // 2)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy.double())
    p0, p1, pw, pb = _leaf(x0), (_leaf(x1) if C1 else None), _leaf(w), _leaf(b)
    out = T.ConvFn.apply(p0, p1, pw, pb, ks, mode)
    out.backward(dy.to(DEV))
    assert _rel(out, ref) < 3e-6
    assert _rel(p0.grad, r0.grad) < 5e-6, "d src0"
    if C1:
        assert _rel(p1.grad, r1.grad) < 5e-6, "d src1"
    assert _rel(pw.grad, rw.grad) < 5e-6, "d weight"
    assert _rel(pb.grad, rb.grad) < 5e-6, "d bias"


@pytest.mark.parametrize("C0,C1,Co,H", [(64, 0, 64, 16), (64, 48, 64, 32), (32, 0, 64, 8)])
def test_resblock_fn_backward(C0, C1, Co, H):
    g = _g(2)
    B, G = 2, 8
    rb = unet_ref.ResBlock(C0 + C1, Co, 32, G).double()
    with torch.no_grad():
        for p in rb.parameters():
            p.copy_(torch.randn(p.shape, generator=g).double() * (0.3 if p.dim() == 1 else 1.0 / math.sqrt(p[0].numel())))
        rb.norm1.weight.add_(1.0), rb.norm2.weight.add_(1.0)
    x0 = torch.randn(B, C0, H, H, generator=g)
    x1 = torch.randn(B, C1, H, H, generator=g) if C1 else None
    temb = torch.randn(B, 32, generator=g)
    vec = torch.randn(B, Co, generator=g)
    r0 = x0.double().requires_grad_(True)
    r1 = x1.double().requires_grad_(True) if C1 else None
    rt = temb.double().requires_grad_(True)
    rv = vec.double().requires_grad_(True)
    ref = rb(torch.cat([r0, r1], 1) if C1 else r0, rt) + rv[:, :, None, None]
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy.double())
    # product: film = Linear(SiLU(temb)) is formed outside the Function
    P = {k: _leaf(v.float()) for k, v in rb.state_dict().items()}
    p0, p1, pt, pv = _leaf(x0), (_leaf(x1) if C1 else None), _leaf(temb), _leaf(vec)
    film = T.LinearFn.apply(T.ActFn.apply(pt, ops.ACT_SILU), P["mlp.weight"], P["mlp.bias"])
    ident = (C0 + C1) == Co
    out = T.ResBlockFn.apply(p0, p1, film, pv, P["conv1.weight"], P["conv1.bias"], P["norm1.weight"], P["norm1.bias"], P["conv2.weight"],
                             P["conv2.bias"], P["norm2.weight"], P["norm2.bias"], None if ident else P["res_conv.weight"],
                             None if ident else P["res_conv.bias"], G, 1e-5)
    out.backward(dy.to(DEV))
    assert _rel(out, ref) < 1e-5
    assert _rel(p0.grad, r0.grad) < 2e-5
    if C1:
        assert _rel(p1.grad, r1.grad) < 2e-5
    assert _rel(pt.grad, rt.grad) < 2e-5, "d temb (through FiLM)"
    assert _rel(pv.grad, rv.grad) < 2e-5
    for k, p in rb.named_parameters():
        assert _rel(P[k].grad, p.grad) < 3e-5, k


@pytest.mark.parametrize("tA,tB", [(False, False), (True, False), (False, True), (True, True)])
def test_bgemm_softmax_fn(tA, tB):
    g = _g(3)
    bt, M, N, K = 3, 37, 50, 29
    A = torch.randn((bt, K, M) if tA else (bt, M, K), generator=g)
    Bm = torch.randn((bt, N, K) if tB else (bt, K, N), generator=g)
    ra, rb = A.double().requires_grad_(True), Bm.double().requires_grad_(True)
    ref = ((ra.transpose(1, 2) if tA else ra) @ (rb.transpose(1, 2) if tB else rb) * 0.3).softmax(-1)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy.double())
    pa, pb = _leaf(A), _leaf(Bm)
    out = T.SoftmaxRowsFn.apply(T.BgemmFn.apply(pa, pb, tA, tB), 0.3)
    out.backward(dy.to(DEV))
    assert _rel(out, ref) < 5e-6 and _rel(pa.grad, ra.grad) < 1e-5 and _rel(pb.grad, rb.grad) < 1e-5


def test_token_side_fns():
    g = _g(4)
    R, K, N = 15, 48, 40
    x = torch.randn(R, K, generator=g)
    w, b = torch.randn(N, K, generator=g) / 7, torch.randn(N, generator=g)
    ga, be = torch.randn(K, generator=g), torch.randn(K, generator=g)
    gm = torch.randn(N, generator=g)
    rx, rw, rbb, rga, rbe, rgm = [t.double().requires_grad_(True) for t in (x, w, b, ga, be, gm)]
    ref = F.gelu(F.linear(F.layer_norm(rx, (K,), rga, rbe, 1e-5), rw, rbb)) * rgm
    ref = ref + F.silu(ref)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy.double())
    px, pw, pb, pga, pbe, pgm = [_leaf(t) for t in (x, w, b, ga, be, gm)]
    y = T.ScaleColsFn.apply(T.ActFn.apply(T.LinearFn.apply(T.LayerNormRowsFn.apply(px, pga, pbe, 1e-5), pw, pb), ops.ACT_GELU), pgm)
    out = T.AddFn.apply(y, T.ActFn.apply(y, ops.ACT_SILU), 1.0)
    out.backward(dy.to(DEV))
    assert _rel(out, ref) < 5e-6
    for p, r in ((px, rx), (pw, rw), (pb, rbb), (pga, rga), (pbe, rbe), (pgm, rgm)):
        assert _rel(p.grad, r.grad) < 2e-5


def test_map_side_fns():
    g = _g(5)
    B, C, H, W = 2, 32, 8, 12
    x = torch.randn(B, C, H, W, generator=g)
    ga, be = torch.randn(C, generator=g), torch.randn(C, generator=g)
    vec = torch.randn(B, C, generator=g)
    idx = torch.tensor([3, 7], dtype=torch.int32)
    rx, rga, rbe, rv = [t.double().requires_grad_(True) for t in (x, ga, be, vec)]
    y = F.layer_norm(rx.permute(0, 2, 3, 1), (C,), rga, rbe, 1e-5).permute(0, 3, 1, 2)
    y = F.normalize(y, dim=1) + rv[:, :, None, None]
    ref = y[torch.arange(B), idx.long()][:, None]
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy.double())
    px, pga, pbe, pv = [_leaf(t) for t in (x, ga, be, vec)]
    out = T.GatherChannelFn.apply(T.AddVecFn.apply(T.ChanNormalizeFn.apply(T.ChanLayerNormFn.apply(px, pga, pbe, 1e-5)), pv), idx.to(DEV))
    out.backward(dy.to(DEV))
    assert _rel(out, ref) < 5e-6
    for p, r in ((px, rx), (pga, rga), (pbe, rbe), (pv, rv)):
        assert _rel(p.grad, r.grad) < 2e-5


def test_losses_and_adam():
    g = _g(6)
    a, b = torch.randn(3, 1, 20, 24, generator=g), torch.randn(3, 1, 20, 24, generator=g)
    slot = torch.zeros(2, device=DEV)
    grad = T.mse_loss_and_grad(a.to(DEV), b.to(DEV), slot[0:1], weight=0.5)
    ra = a.double().requires_grad_(True)
    l = F.mse_loss(ra, b.double())
    (0.5 * l).backward()
    assert abs(float(slot[0]) - float(l)) < 1e-6 * float(l) + 1e-9 and _rel(grad, ra.grad) < 1e-6
    lab = torch.randn(2, 1, 32, 48, generator=g)
    for oh, ow in ((16, 24), (8, 12), (4, 6)):
        ref = F.interpolate(lab.double(), size=(oh, ow), mode="bilinear", align_corners=False, antialias=False)
        assert _rel(T.resize_bilinear(lab.to(DEV), oh, ow), ref) < 2e-6
    # Adam == torch.optim.Adam (L2-in-grad weight decay)
    p0 = torch.randn(1000, generator=g)
    ref_p = p0.clone().requires_grad_(True)
    ropt = torch.optim.Adam([ref_p], lr=2e-3, betas=(0.9, 0.99), weight_decay=1e-2)
    pp = nn.Parameter(p0.clone().to(DEV))
    opt = T.FusedAdam([pp], lr=2e-3, betas=(0.9, 0.99), weight_decay=1e-2)
    for it in range(3):
        gr = torch.randn(1000, generator=g)
        ref_p.grad = gr.clone()
        ropt.step()
        opt.zero_grad()          # (p.grad = None: autograd would now hand its gradient tensor over; here the test does)
        pp.grad = gr.to(DEV)
        opt.step()
    assert _rel(pp.data, ref_p.data) < 1e-6


def test_gather_segments_fills_the_flat_gradient_buffer_in_one_launch():
    """idiff_gather_segments through FusedAdam._collect: gradients that autograd left as separate tensors (contiguous, a strided view,
    a parameter without gradient, one whose gradient already IS its slice of the flat buffer) end up in the flat buffer bit for bit,
    zeros where there was none; segment sizes around the 4096-element block of the kernel."""
    g = _g(61)
    sizes = [1, 7, 4095, 4096, 4097, 3 * 4096 + 5, 50000, 2]
    ps = [nn.Parameter(torch.randn(n, generator=g).to(DEV)) for n in sizes]
    opt = T.FusedAdam(ps, lr=1e-3)
    opt.zero_grad()
    want = []
    for i, (pp, n) in enumerate(zip(ps, sizes)):
        if i == 1:
            want.append(torch.zeros(n))            # no gradient this step
            continue
        gr = torch.randn(n, generator=g)
        want.append(gr)
        if i == 3:
            big = torch.zeros(2 * n, device=DEV)
            big[::2] = gr.to(DEV)
            pp.grad = big[::2]                     # not contiguous: copied first
        else:
            pp.grad = gr.to(DEV)
    flats = opt.flat_grads()
    flat = flats[0] if isinstance(flats, (list, tuple)) else flats
    assert torch.equal(flat.cpu(), torch.cat(want))
    for pp, w in zip(ps, want):                    # p.grad is now the parameter's view of the flat buffer
        assert torch.equal(pp.grad.cpu(), w)
    opt.flat_grads()                               # nothing left to gather: a second call changes nothing
    assert torch.equal(flat.cpu(), torch.cat(want))


@pytest.mark.parametrize("R,K,N,strided", [(5, 29, 50, False), (37, 256, 70, True), (160, 256, 256, False), (160, 64, 1024, True), (33, 512, 64, False)])
def test_linear_mfma_fwd_vs_fp64(R, K, N, strided):
    """idiff_linear_mfma_fwd (LinearFn's forward): y = x W^T + b on the matrix cores for any row count; x may be a column slice."""
    g = _g(62)
    xb = torch.randn(R, K + (8 if strided else 0), generator=g)
    x = xb[:, 4:4 + K] if strided else xb
    w, b = torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    ref = x.double() @ w.double().t() + b.double()
    xd = xb.to(DEV)
    y = T.LinearFn.apply(xd[:, 4:4 + K] if strided else xd, w.to(DEV), b.to(DEV))
    assert _rel(y, ref) < 3e-6
    y0 = T.LinearFn.apply(xd[:, 4:4 + K] if strided else xd, w.to(DEV), None)
    assert _rel(y0, ref - b.double()) < 3e-6


def _oracle_pair(model):
    opt = pipeline.load_options()
    mo = opt['models']['DriftNoise']
    refs = []
    for key, net in (('dnet_settings', model.drift_net), ('nnet_settings', model.noise_net)):
        s = {k: v for k, v in dict(mo[key]).items() if k not in ("module_name", "class_name")}
        smm = nn.ModuleList([unet_ref.ScoreMapModule(visual_dim=mo['score_map_ngf'] * m) for m in mo['score_map_ch_mult']])
        r = unet_ref.LearnableForwardUNet_MultiScoreMap(CLIP_ScoreMapModule=smm, use_image_context=True, **s)
        r.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()})
        refs.append(r)
    return refs


def test_unet_param_gradients_and_train_steps_vs_oracle():
    B, H, T_ = 2, 32, 20
    model, sde = pipeline.build(phase="train", device=torch.device(DEV), T=T_, seed=0, score_map_dropout=0.0)
    model.set_train()
    with torch.no_grad():  # make the tiny-gamma / zero-bias paths carry signal
        for net in (model.drift_net, model.noise_net):
            for m in net.CLIP_ScoreMapModule:
                m.gamma.fill_(0.3)
    rd, rn = _oracle_pair(model)
    te = unet_ref.StubTextEncoder()
    batch = make_batch(B, H, seed=3)
    g = _g(7)
    t = torch.tensor([[[[5]]], [[[17]]]])
    eps = torch.randn(batch['input'].shape, generator=g)
    # ---- oracle step (torch autograd + torch Adam on the CPU) ----
    osde = sde_ref.DriftSDERef(T_, rd, rn, max_sigma=0.4)
    _, x_t, _, std_noise, _ = osde.forward_diffusion(batch['target'], batch['input'], t, eps)
    opt_d = torch.optim.Adam(rd.parameters(), lr=2e-5, betas=(0.9, 0.99), weight_decay=1e-4)
    opt_n = torch.optim.Adam(rn.parameters(), lr=2e-5, betas=(0.9, 0.99), weight_decay=1e-4)

    def oracle_loss():
        tt = t.reshape(-1)
        pd, dsm = rd(x_t - batch['input'], batch['input'], tt, batch['names'], te, image_context=batch['A_emb'])
        pn, nsm = rn(x_t - batch['input'], x_t, tt, batch['names'], te, image_context=batch['A_emb'])
        tgt = batch['input'] - batch['target']

        def pyr(sms, lab):
            tot = 0
            for i, sm in enumerate(sms):
                lb = lab if i == 0 else F.interpolate(lab, size=(H >> i, H >> i), mode="bilinear", align_corners=False, antialias=False)
                tot = tot + F.mse_loss(sm, lb)
            return tot / 2.0
        return F.mse_loss(pd, tgt) + F.mse_loss(pn, std_noise) + pyr(dsm, tgt) + pyr(nsm, std_noise)

    opt_d.zero_grad(), opt_n.zero_grad()
    l0 = oracle_loss()
    l0.backward()
    ref_grads = {("d", k): p.grad.clone() for k, p in rd.named_parameters()}
    ref_grads.update({("n", k): p.grad.clone() for k, p in rn.named_parameters()})
    opt_d.step(), opt_n.step()
    # ---- product step ----
    model.input = batch['input'].to(DEV)
    model.target = batch['target'].to(DEV)
    model.names = batch['names']
    model.A_emb = batch['A_emb'].to(DEV)
    model.t, model.drift_noised_x, _, model.std_noise, _ = sde.forward_diffusion(model.target, model.input, t=t, eps=eps.to(DEV))
    assert _rel(model.drift_noised_x, x_t) < 1e-6
    loss, _ = model.optimize_parameters()
    assert abs(loss - float(l0)) < 2e-5 * abs(float(l0)), (loss, float(l0))
    # gradients live in the optimizers' flat buffers, still intact after step()
    worst = 0.0
    for tag, net in (("d", model.drift_net), ("n", model.noise_net)):
        for k, p in net.named_parameters():
            r = ref_grads[(tag, k)]
            scale = float(r.abs().max())
            if scale < 1e-12:
                assert float(p.grad.abs().max()) < 1e-9, (tag, k)
                continue
            e = float((p.grad.cpu() - r).abs().max()) / scale
            worst = max(worst, e)
            assert e < 2e-3, (tag, k, e)
    print("worst relative parameter-gradient error", worst)
    # parameters after one Adam step
    # (the first Adam step is -lr*g/(|g|+eps) ~ -lr*sign(g): elements whose gradient is ~0 may legitimately differ by up
    #  to 2*lr, so the check is on the bulk: >= 99 % of all elements within 0.1*lr, none beyond 2.5*lr)
    lr, n_all, n_bad = 2e-5, 0, 0
    for net, ref in ((model.drift_net, rd), (model.noise_net, rn)):
        for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
            d = (p.detach().cpu() - q.detach()).abs()
            assert float(d.max()) < 2.5 * lr, k
            n_all += d.numel()
            n_bad += int((d > 0.1 * lr).sum())
    assert n_bad < 0.01 * n_all, (n_bad, n_all)
    # second step: loss is finite and the inference path sees the updated weights (prepared-weight cache invalidated)
    loss2, _ = model.optimize_parameters()
    assert math.isfinite(loss2)
    model.set_eval()
    with torch.no_grad():
        p_inf = model.drift_net(model.drift_noised_x - model.input, model.input, t.reshape(-1).to(DEV), model.names, model.text_encoder,
                                image_context=model.A_emb)[0]
    opt_d.zero_grad(), opt_n.zero_grad()
    oracle_loss().backward()
    opt_d.step(), opt_n.step()
    with torch.no_grad():
        p_ref = rd.eval()(x_t - batch['input'], batch['input'], t.reshape(-1), batch['names'], te, image_context=batch['A_emb'])[0]
    assert _rel(p_inf, p_ref) < 1e-3


# ---------------------------------------------------------------------------------------------------
# Winograd weight gradient (conv_wino_wgrad.hip): every eligible gather variant against fp64 autograd
@pytest.mark.parametrize("B,C0,C1,Cout,H,W,variant", [
    (2, 64, 0, 64, 16, 16, "plain"),
    (3, 80, 0, 128, 8, 48, "plain"),          # partial last 64-channel block, several samples per split
    (2, 64, 80, 64, 32, 32, "concat"),        # second source starts on a 64-channel block boundary
    (2, 32, 0, 64, 16, 32, "prologue"),
    (2, 32, 0, 64, 8, 16, "upsample"),        # out 16x32
    (1, 16, 0, 64, 2, 16, "plain"),           # a single chunk: every edge at once
    # F(4x4,3x3) form (output tiles by 4 x 16): blocks of 64 co x 32 ci
    (2, 64, 0, 64, 16, 32, "plain"),
    (1, 16, 0, 64, 4, 16, "plain"),           # a single chunk: every edge at once, half a channel block
    (3, 48, 0, 128, 8, 48, "prologue"),       # partial last 32-channel block, two co blocks
    (2, 64, 80, 64, 32, 32, "concat"),        # second source starts on a block boundary; 144 = 4.5 blocks
    (5, 32, 0, 64, 64, 64, "plain"),          # many chunks per split
    (3, 48, 0, 128, 8, 24, "upsample"),       # out 16x48: every border chunk kind, partial last channel block
    (2, 32, 0, 64, 2, 8, "upsample"),         # out 4x16: a single chunk per sample
    (2, 32, 0, 64, 6, 16, "upsample"),        # out 12x32: F(4x4) (12 % 4 == 0)
    (2, 32, 0, 64, 5, 16, "upsample"),        # out 10x32: not tiled by 4 -> F(2x2)
])
def test_conv_wgrad_winograd_vs_fp64(B, C0, C1, Cout, H, W, variant):
    g = _g(41)
    Cin = C0 + C1
    x0 = torch.randn(B, C0, H, W, generator=g)
    x1 = torch.randn(B, C1, H, W, generator=g) if C1 else None
    xin = x0.double() if x1 is None else torch.cat([x0, x1], 1).double()
    pro, mode = None, ops.CONV_NORMAL
    if variant == "prologue":
        pa, pb = torch.randn(B, C0, generator=g), torch.randn(B, C0, generator=g)
        xin = xin * pa.double()[:, :, None, None] + pb.double()[:, :, None, None]
        xin = xin / (1 + torch.exp(-xin))
        pro = (pa.to(DEV), pb.to(DEV))
    if variant == "upsample":
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
        mode = ops.CONV_UPSAMPLE2
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(xin, w, None, padding=1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    dw = T.conv2d_wgrad(x0.to(DEV), None if x1 is None else x1.to(DEV), mode, 3, dy.to(DEV), Cin, pro=pro)
    # F(4x4,3x3) form (algo 3) where the output tiles by 4 x 16 (normal and upsample mode), else the F(2x2,3x3) form (algo 1)
    want = 3 if (y.shape[2] % 4 == 0 and y.shape[3] % 16 == 0 and (x1 is None or C0 % 32 == 0)) else 1
    assert ops._lib.load().idiff_conv2d_wgrad_last_algo() == want, "the expected Winograd weight-gradient kernel did not run"
    assert _rel(dw, w.grad) < (2e-5 if want == 3 else 1e-5), "winograd wgrad"
    # accumulate form
    dw2 = T.conv2d_wgrad(x0.to(DEV), None if x1 is None else x1.to(DEV), mode, 3, dy.to(DEV), Cin, pro=pro, dw=dw.clone(), accumulate=True)
    assert _rel(dw2, 2 * w.grad) < 2e-5


@pytest.mark.parametrize("B,C0,C1,Cout,H,W", [(2, 64, 80, 64, 16, 16), (3, 128, 0, 128, 8, 16), (1, 48, 0, 64, 16, 32), (2, 256, 0, 192, 16, 8)])
def test_conv_wgrad_streaming_1x1_vs_fp64(B, C0, C1, Cout, H, W):
    g = _g(42)
    Cin = C0 + C1
    x0 = torch.randn(B, C0, H, W, generator=g)
    x1 = torch.randn(B, C1, H, W, generator=g) if C1 else None
    xin = x0.double() if x1 is None else torch.cat([x0, x1], 1).double()
    dy = torch.randn(B, Cout, H, W, generator=g)
    ref = torch.einsum("bohw,bihw->oi", dy.double(), xin).reshape(Cout, Cin, 1, 1)
    dw = T.conv2d_wgrad(x0.to(DEV), None if x1 is None else x1.to(DEV), ops.CONV_NORMAL, 1, dy.to(DEV), Cin)
    assert ops._lib.load().idiff_conv2d_wgrad_last_algo() == 2, "the streaming 1x1 weight-gradient kernel did not run"
    assert _rel(dw, ref) < 5e-6


@pytest.mark.parametrize("B,C0,Cout,H,W", [(2, 16, 64, 16, 32), (3, 64, 128, 32, 16), (1, 8, 64, 16, 64), (2, 24, 64, 64, 8)])
def test_conv_wgrad_streaming_1x1_of_the_pixel_unshuffle_vs_fp64(B, C0, Cout, H, W):
    """The downsample layers (pixel-unshuffle(2) + 1x1) on the streaming kernel: row pairs of virtual channels de-interleaved in registers;
    whole and partial 64-row blocks of virtual channels, runs of 128 output pixels that span 1 / 2 / 4 / 32 output rows."""
    g = _g(43)
    x0 = torch.randn(B, C0, H, W, generator=g)
    dy = torch.randn(B, Cout, H // 2, W // 2, generator=g)
    ref = torch.einsum("bohw,bihw->oi", dy.double(), F.pixel_unshuffle(x0.double(), 2)).reshape(Cout, 4 * C0, 1, 1)
    dw = T.conv2d_wgrad(x0.to(DEV), None, ops.CONV_UNSHUFFLE2, 1, dy.to(DEV), 4 * C0)
    assert ops._lib.load().idiff_conv2d_wgrad_last_algo() == 2, "the streaming 1x1 weight-gradient kernel did not run"
    assert _rel(dw, ref) < 5e-6


def test_conv_wgrad_winograd_random_shapes_vs_fp64():
    rng = __import__("numpy").random.RandomState(9)
    lib = ops._lib.load()
    for case in range(10):
        B = int(rng.randint(1, 4))
        two = case % 3 == 1
        C0 = 64 * int(rng.randint(1, 3)) if two else 16 * int(rng.randint(1, 9))
        C1 = 16 * int(rng.randint(1, 5)) if two else 0
        Cout = 64 * int(rng.randint(1, 3))
        H, W = 2 * int(rng.randint(1, 9)), 16 * int(rng.randint(1, 4))
        g = _g(200 + case)
        x0 = torch.randn(B, C0, H, W, generator=g)
        x1 = torch.randn(B, C1, H, W, generator=g) if two else None
        xin = x0.double() if x1 is None else torch.cat([x0, x1], 1).double()
        w = torch.zeros(Cout, C0 + C1, 3, 3, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(xin, w, None, padding=1)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy.double())
        dw = T.conv2d_wgrad(x0.to(DEV), None if x1 is None else x1.to(DEV), ops.CONV_NORMAL, 3, dy.to(DEV), C0 + C1)
        want = 3 if (H % 4 == 0 and (not two or C0 % 32 == 0)) else 1
        assert lib.idiff_conv2d_wgrad_last_algo() == want, (case, B, C0, C1, Cout, H, W)
        assert _rel(dw, w.grad) < 2e-5, (case, B, C0, C1, Cout, H, W)


def test_optimize_score_map_by_name_with_the_reference_size_argument():
    """CLIPDriftModel.optimize_score_map(score_maps, label, size=[224, 224]) exactly as the reference calls it
    (models/drift_noise_model.py:234-240, 290-291): sum_i MSE(sm_i, Resize(224 // m_i)(label)) / 2, m = [1, 2, 4, 8]; torchvision-0.14
    tensor Resize = bilinear without antialiasing.  Value and per-map gradients against torch."""
    import torch.nn.functional as F
    from instancediff_amd import pipeline
    model, _ = pipeline.build(phase="train", device=torch.device(DEV), T=10, seed=0)
    g = torch.Generator().manual_seed(61)
    B, S = 3, 224
    label = torch.randn(B, 1, S, S, generator=g)
    sms = [torch.randn(B, 1, S // m, S // m, generator=g) for m in (1, 2, 4, 8)]
    want = 0.0
    leaves = [s.clone().requires_grad_(True) for s in sms]
    for m, sm in zip((1, 2, 4, 8), leaves):
        lb = label if m == 1 else F.interpolate(label, size=(S // m, S // m), mode="bilinear", align_corners=False, antialias=False)
        want = want + F.mse_loss(sm, lb)
    want = want / 2.0
    want.backward()
    loss, grads = model.optimize_score_map([s.to(DEV) for s in sms], label.to(DEV), size=[224, 224], want_grads=True)
    assert abs(float(loss) - float(want)) < 2e-6 * abs(float(want)), (float(loss), float(want))
    for gh, leaf in zip(grads, leaves):
        assert float((gh.cpu() - leaf.grad).abs().max()) < 2e-6 * float(leaf.grad.abs().max()) + 1e-12  # the /2 of :240 included
    # default size = the label's own
    assert float(model.optimize_score_map([s.to(DEV) for s in sms], label.to(DEV))) == float(loss)


def test_dropout_kernel_mask_is_a_function_of_seed_and_offset():
    """idiff_dropout: kept iff (w >> 8) * 2^-24 >= p on the Philox words of (seed, offset + i / 4); kept values scaled by 1 / (1 - p);
    the backward is the same call (same mask); p = 0 keeps everything."""
    n, p, seed, off = 80 * 1024 + 3, 0.1, 77, 1234
    g = _g(81)
    x = torch.randn(n, generator=g).to(DEV)
    y = ops.dropout(x, p, seed, off)
    words = ops.philox_raw((n + 3) // 4, DEV, seed, off).reshape(-1)[:n].to(torch.int64) & 0xFFFFFFFF
    keep = ((words >> 8).double() * 2.0 ** -24) >= p
    assert torch.equal(y != 0, keep & (x != 0))
    assert torch.equal(y[keep], x[keep] * torch.tensor(1.0 / (1.0 - p), dtype=torch.float32, device=DEV))
    frac = float(keep.float().mean())
    assert abs(frac - 0.9) < 5e-3, frac
    assert torch.equal(ops.dropout(x, p, seed, off), y) and not torch.equal(ops.dropout(x, p, seed, off + 1), y)
    assert torch.equal(ops.dropout(x, 0.0, seed, off), x)
    # autograd Function: the backward regenerates the mask of the forward
    T.DropoutState.reset(5)
    xr = x.clone().requires_grad_(True)
    out = T.dropout(xr, p, True)
    out.backward(torch.ones_like(out))
    assert torch.equal(xr.grad != 0, out != 0) or float(((xr.grad != 0) ^ (out != 0)).sum()) <= float((x == 0).sum())
    assert T.dropout(xr, p, False) is xr and T.dropout(xr, 0.0, True) is xr


def test_training_dropout_of_the_decoder_blocks_vs_oracle_with_the_same_masks():
    """set_train() trains the function the reference trains: TransformerDecoderLayer(dropout=0.1) -- Attention.proj_drop on both
    attentions, the MLP's inner Dropout and the block's Dropout (models/_modified_BiomedCLIP.py:448-478,520-549).  The product's
    masks (Philox, regenerated in the backward) are recorded in call order and injected into the oracle's autograd: loss and every
    parameter gradient must agree; eval() is untouched by the option."""
    B, H, T_ = 2, 32, 20
    model, sde = pipeline.build(phase="train", device=torch.device(DEV), T=T_, seed=0)   # default: the reference's 0.1
    assert all(m.context_decoder.dropout == 0.1 for m in model.drift_net.CLIP_ScoreMapModule)
    model.set_train()
    with torch.no_grad():
        for net in (model.drift_net, model.noise_net):
            for m in net.CLIP_ScoreMapModule:
                m.gamma.fill_(0.3)
    rd, rn = _oracle_pair(model)
    rd.train(), rn.train()
    te = unet_ref.StubTextEncoder()
    batch = make_batch(B, H, seed=13)
    g = _g(17)
    t = torch.tensor([[[[5]]], [[[17]]]])
    eps = torch.randn(batch['input'].shape, generator=g)
    model.input, model.target = batch['input'].to(DEV), batch['target'].to(DEV)
    model.names, model.A_emb = batch['names'], batch['A_emb'].to(DEV)
    model.t, model.drift_noised_x, _, model.std_noise, _ = sde.forward_diffusion(model.target, model.input, t=t, eps=eps.to(DEV))
    T.DropoutState.reset(4242)
    T.DropoutState.record = []
    try:
        rec, _, _, _ = T.forward_backward_inputRes(model)
        record = T.DropoutState.record
    finally:
        T.DropoutState.record = None
    n_sites = 2 * 4 * 3 * 4   # nets x ScoreMapModules x decoder layers x dropout sites
    r = rec.cpu()
    loss = float(r[0] + r[1] + r[2:6].sum() / 2 + r[6:10].sum() / 2)
    masks = [(ops.dropout(torch.ones(shape, device=DEV), p, seed, off) != 0).float().cpu() for shape, seed, off, p in record]
    if len(record) == n_sites // 4:
        # stacked token chains (r05): one draw per (net, layer, site) over the [L = 4 levels, rows, width] stack.  The oracle calls its
        # dropouts level by level (drift net then noise net; level, layer, site): deal the level slices out in that order
        assert all(len(shape) == 3 and shape[0] == 4 for shape, _, _, _ in record)
        per_net = len(masks) // 2
        queue = []
        for net_i in range(2):
            mine = masks[net_i * per_net:(net_i + 1) * per_net]   # [layer][site]
            for level in range(4):
                for j in range(per_net):
                    queue.append(mine[j][level])
        masks = queue
    assert len(masks) == n_sites, len(masks)
    # the same masks, as 0/1 tensors, in the oracle's call order (drift net then noise net; level, layer, site)
    unet_ref.InjectedDropout.queue = masks
    kept = sum(float(m.sum()) for m in unet_ref.InjectedDropout.queue) / sum(m.numel() for m in unet_ref.InjectedDropout.queue)
    assert 0.88 < kept < 0.92, kept
    osde = sde_ref.DriftSDERef(T_, rd, rn, max_sigma=0.4)
    _, x_t, _, std_noise, _ = osde.forward_diffusion(batch['target'], batch['input'], t, eps)
    tt = t.reshape(-1)
    pd, dsm = rd(x_t - batch['input'], batch['input'], tt, batch['names'], te, image_context=batch['A_emb'])
    pn, nsm = rn(x_t - batch['input'], x_t, tt, batch['names'], te, image_context=batch['A_emb'])
    assert not unet_ref.InjectedDropout.queue, "the oracle consumed a different number of dropout masks"
    tgt = batch['input'] - batch['target']

    def pyr(sms, lab):
        tot = 0
        for i, sm in enumerate(sms):
            lb = lab if i == 0 else F.interpolate(lab, size=(H >> i, H >> i), mode="bilinear", align_corners=False, antialias=False)
            tot = tot + F.mse_loss(sm, lb)
        return tot / 2.0
    l0 = F.mse_loss(pd, tgt) + F.mse_loss(pn, std_noise) + pyr(dsm, tgt) + pyr(nsm, std_noise)
    l0.backward()
    assert abs(loss - float(l0.detach())) < 2e-5 * abs(float(l0.detach())), (loss, float(l0.detach()))
    worst = 0.0
    for net, ref in ((model.drift_net, rd), (model.noise_net, rn)):
        refg = dict(ref.named_parameters())
        for k, p_ in net.named_parameters():
            rg = refg[k].grad
            scale = float(rg.abs().max())
            if scale < 1e-12:
                assert float(p_.grad.abs().max()) < 1e-9, k
                continue
            e = float((p_.grad.cpu() - rg).abs().max()) / scale
            worst = max(worst, e)
            assert e < 2e-3, (k, e)
    print(f"decoder dropout 0.1: {n_sites} masks, kept {kept:.4f}; loss {loss:.6f} (oracle {float(l0.detach()):.6f}), worst relative gradient error {worst:.2e}")
    # a second backward with other masks gives other gradients (the masks matter), eval() ignores the option
    g1 = torch.cat([p_.grad.reshape(-1) for p_ in model.drift_net.CLIP_ScoreMapModule.parameters() if p_.grad is not None]).clone()
    T.forward_backward_inputRes(model)
    g2 = torch.cat([p_.grad.reshape(-1) for p_ in model.drift_net.CLIP_ScoreMapModule.parameters() if p_.grad is not None])
    assert not torch.equal(g1, g2)


def test_training_gradients_are_the_same_bits_from_run_to_run():
    """Every reduction of the backward is a fixed-order one (split partials of the weight gradients, GroupNorm / LayerNorm plane sums,
    key splits of the cross-attention, the gather into the flat buffers) and the two nets' streams share no tensor: two
    forward+backward passes from the same state and batch give bit-identical loss records and flat gradient buffers (128x128, B = 3, so
    the F(4x4,3x3) weight-gradient kernel, its split reduce and the fused cross-attention backward all run with several splits)."""
    B, H, T_ = 3, 128, 20
    model, sde = pipeline.build(phase="train", device=torch.device(DEV), T=T_, seed=0, score_map_dropout=0.0)
    model.set_train()
    with torch.no_grad():
        for net in (model.drift_net, model.noise_net):
            for m in net.CLIP_ScoreMapModule:
                m.gamma.fill_(0.3)
    batch = make_batch(B, H, seed=5)
    g = _g(70)
    t = torch.tensor([3, 11, 17]).reshape(B, 1, 1, 1)
    eps = torch.randn(batch['input'].shape, generator=g)
    model.input, model.target = batch['input'].to(DEV), batch['target'].to(DEV)
    model.names, model.A_emb = batch['names'], batch['A_emb'].to(DEV)
    model.t, model.drift_noised_x, _, model.std_noise, _ = sde.forward_diffusion(model.target, model.input, t=t, eps=eps.to(DEV))
    runs = []
    for _ in range(2):
        rec = T.forward_backward_inputRes(model)[0]
        flats = [f.clone() for f in model.drift_optimizer.flat_grads() + model.noise_optimizer.flat_grads()]
        runs.append((rec.clone(), flats))
    torch.cuda.synchronize()
    assert torch.equal(runs[0][0], runs[1][0]), "loss records differ from run to run"
    assert len(runs[0][1]) == len(runs[1][1]) > 0
    for a, b in zip(runs[0][1], runs[1][1]):
        assert float(a.abs().max()) > 0
        assert torch.equal(a, b), "flat gradient buffers differ from run to run"


@pytest.mark.parametrize("B,R,N,Cm", [(2, 20, 1024, 256), (1, 20, 4096 + 32, 256), (3, 7, 96, 256), (2, 32, 65536, 256),
                                      (2, 20, 1024, 72), (1, 20, 4096 + 32, 72), (3, 7, 96, 72), (2, 20, 65536, 72)])
def test_fused_scoremap_cross_attention_forward_backward_vs_fp64(B, R, N, Cm):
    """SmmXattnFn (training path): o and lse of the flash-decoding forward, dqf and dmem of the one-pass fused backward, against
    fp64 autograd of softmax(scale * qf mem) mem^T; the key split of the big case covers 64 splits with the fixed-order combine.
    Cm = 72: the compact (C + 1)-row memory of the 64-channel levels (r05), padding rows zero as the memory projection leaves them."""
    g = _g(90 + R)
    qf = (torch.randn(B, R, Cm, generator=g) * 0.3)
    mem = torch.randn(B, Cm, N, generator=g)
    do = torch.randn(B, R, Cm, generator=g)
    if Cm == 72:
        mem[:, 65:] = 0.0
    scale = 0.125
    q64, m64 = qf.double().requires_grad_(True), mem.double().requires_grad_(True)
    s = torch.einsum('brc,bcn->brn', q64, m64) * scale
    o64 = torch.einsum('brn,bcn->brc', s.softmax(-1), m64)
    o64.backward(do.double())
    qd, md = qf.to(DEV).requires_grad_(True), mem.to(DEV).requires_grad_(True)
    o = T.SmmXattnFn.apply(qd, md, scale)
    o.backward(do.to(DEV))
    for name, got, ref, tol in (("o", o, o64, 2e-5), ("dqf", qd.grad, q64.grad, 5e-5), ("dmem", md.grad, m64.grad, 5e-5)):
        e = float((got.detach().cpu().double() - ref.detach()).abs().max() / ref.detach().abs().max())
        print(f"fused cross-attention B={B} R={R} N={N} Cm={Cm}: {name} rel err {e:.2e}")
        assert e < tol, (name, e)


@pytest.mark.parametrize("C,H", [(64, 16), (256, 8)])
def test_scaled_decoder_scoremap_module_gradients_vs_fp64_oracle(C, H):
    """ScoreMapModule over ContextDecoder_Hierachical / TransformerDecoderLayer_scaled (models/_modified_BiomedCLIP.py:552-590,
    1247-1308): training-mode forward (dropout 0) and every parameter gradient, the branch gains gamma_sa/ca/mlp included, against
    the fp64 oracle module with the same weights."""
    from instancediff_amd.models.modules.MSM_degEmb_Unet import ScoreMapModule
    from instancediff_amd.models.modules.unet_autograd import _smm
    B, K = 2, 5
    g = _g(31)
    smm = ScoreMapModule(visual_dim=C, decoder_layers=2, decoder_type="ContextDecoder_Hierachical", dropout=0.0).to(DEV).train()
    with torch.no_grad():
        smm.gamma.fill_(0.3)
        for l in smm.context_decoder.decoder:
            for name in ("gamma_sa", "gamma_ca", "gamma_mlp"):
                getattr(l, name).copy_((0.1 + 0.2 * torch.randn((1, 1, 256), generator=g)).to(DEV))
    ref = unet_ref.ScoreMapModule(visual_dim=C, decoder_layers=2, decoder_type="ContextDecoder_Hierachical")
    ref.load_state_dict({k: v.detach().cpu() for k, v in smm.state_dict().items()})
    ref = ref.double()
    from instancediff_amd.models.text_encoder import StubTextEncoder
    te = StubTextEncoder().to(DEV)
    feat = torch.randn(B, C, H, H, generator=g)
    wgt = torch.randn(B, K, H, H, generator=g)
    idx = torch.tensor([1, 3], dtype=torch.int32, device=DEV)
    fr = feat.double().requires_grad_(True)
    out_r = ref(fr, unet_ref.StubTextEncoder().double())
    (out_r * wgt.double()).sum().backward()
    fd = feat.to(DEV).requires_grad_(True)
    score, _ = _smm(smm, fd, te, idx)
    (score * wgt.to(DEV)).sum().backward()
    assert _rel(score, out_r.float()) < 2e-5
    assert _rel(fd.grad, fr.grad.float()) < 5e-4
    rg = dict(ref.named_parameters())
    seen = 0
    for k, p in smm.named_parameters():
        r = rg[k].grad
        assert p.grad is not None, k
        scale = float(r.abs().max())
        e = float((p.grad.cpu().double().reshape(r.shape) - r).abs().max()) / max(scale, 1e-12)
        assert e < 1e-3, (k, e)
        seen += "gamma_" in k
    assert seen == 6


def test_fork_sums_the_consumers_gradients_in_one_launch_and_skipcat_shares_a_buffer():
    """train_ops.fork / SkipCatFn (r05): a feature map with several consumers hands out aliases, the consumers' gradients -- one of them
    a channel slice of a bigger tensor -- meet in idiff_sum_n; a skip cat(x, emb) whose two producers wrote into one buffer is handed
    out without a copy and routes the gradient slices back.  Against torch autograd of the same graph on the CPU."""
    g = _g(31)
    B, C, Ce, H, W = 3, 8, 4, 8, 12
    a = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(B, C + Ce, H, W, generator=g)
    ad = _leaf(a)
    buf = torch.empty(B, C + Ce, H, W, device=DEV)
    x = T.AddFn.apply(ad, ad, 1.0)                       # 2a (a library Function output)
    buf[:, :C].copy_(x.detach())
    xs = T.ForkFn.apply(x, 3)
    e = T.AddFn.apply(xs[1][:, :Ce].contiguous(), xs[1][:, :Ce].contiguous(), 0.5)   # 1.5 * x[:, :Ce]
    buf[:, C:].copy_(e.detach())

    class _Alias(torch.autograd.Function):               # stands in for a producer that wrote into its slice of buf
        @staticmethod
        def forward(ctx, t, slot):
            return slot.t

        @staticmethod
        def backward(ctx, gr):
            return gr, None
    skip = T.SkipCatFn.apply(_Alias.apply(xs[2], T._Slot(buf[:, :C].detach())), _Alias.apply(e, T._Slot(buf[:, C:].detach())), T._Slot(buf))
    wd = w.to(DEV)
    loss = (skip * wd).sum() + (xs[0] * xs[0]).sum()
    loss.backward()
    ar = a.clone().requires_grad_(True)
    xr = 2 * ar
    er = 1.5 * xr[:, :Ce]
    ((torch.cat([xr, er], 1) * w).sum() + (xr * xr).sum()).backward()
    assert _rel(ad.grad, ar.grad) < 1e-6
    # sum_n: 2..4 operands, one of them batch-strided
    ts = [torch.randn(B, C, H, W, generator=g).to(DEV) for _ in range(3)] + [torch.randn(B, C + 4, H, W, generator=g).to(DEV)[:, 4:]]
    for n in (2, 3, 4):
        ref = ts[-1].clone()
        for t in ts[:n - 1]:
            ref = ref + t
        assert _rel(T.sum_n([ts[-1]] + ts[:n - 1]), ref) < 1e-6


def test_token_side_fused_functions_vs_torch_autograd():
    """HeadFoldFn (per-head k / v folds with the head dimension in the GEMM strides), Linear3Fn (packed q | k | v projection) and
    TokenAttnFn (few-token self-attention, one fused backward launch) against torch autograd of the plain formulas in fp64."""
    g = _g(41)
    B, K, heads, Wd = 3, 5, 4, 256
    dh, R = Wd // heads, B * K
    # HeadFoldFn
    for fold in ("in", "out"):
        x = torch.randn(R, Wd if fold == "in" else heads * Wd, generator=g) * 0.5
        w = torch.randn(Wd, Wd, generator=g) * 0.1
        xd, wd = _leaf(x), _leaf(w)
        y = T.HeadFoldFn.apply(xd if fold == "in" else xd.reshape(R, heads, Wd), wd, heads, fold)
        up = torch.randn(y.shape, generator=g)
        (y * up.to(DEV)).sum().backward()
        xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
        if fold == "in":
            yr = torch.einsum("rhd,hdn->rhn", xr.reshape(R, heads, dh), wr.reshape(heads, dh, Wd))
        else:
            yr = torch.einsum("rhn,hdn->rhd", xr.reshape(R, heads, Wd), wr.reshape(heads, dh, Wd)).reshape(R, Wd)
        (yr * up.double()).sum().backward()
        assert _rel(y, yr) < 2e-6 and _rel(xd.grad, xr.grad.reshape(x.shape)) < 2e-6 and _rel(wd.grad, wr.grad) < 2e-6, fold
    # Linear3Fn + TokenAttnFn
    x = torch.randn(R, Wd, generator=g)
    ws = [torch.randn(Wd, Wd, generator=g) * 0.08 for _ in range(3)]
    xd, wds = _leaf(x), [_leaf(w) for w in ws]
    qkv = T.Linear3Fn.apply(xd, *wds)
    a = T.TokenAttnFn.apply(qkv, B, K, heads, dh ** -0.5)
    up = torch.randn(R, Wd, generator=g)
    (a * up.to(DEV)).sum().backward()
    xr, wrs = x.double().requires_grad_(True), [w.double().requires_grad_(True) for w in ws]
    q, k, v = [(xr @ w.t()).reshape(B, K, heads, dh).permute(0, 2, 1, 3) for w in wrs]
    p = torch.softmax(q @ k.transpose(-1, -2) * dh ** -0.5, dim=-1)
    ar = (p @ v).permute(0, 2, 1, 3).reshape(R, Wd)
    (ar * up.double()).sum().backward()
    assert _rel(a, ar) < 3e-6 and _rel(xd.grad, xr.grad) < 1e-5
    for wd_, wr_ in zip(wds, wrs):
        assert _rel(wd_.grad, wr_.grad) < 1e-5


@pytest.mark.parametrize("B,H,W", [(2, 16, 16), (1, 24, 40), (3, 8, 8)])
def test_compact_memory_function_forward_backward_vs_fp64(B, H, W):
    """CompactMemFn (r05: the (C + 1)-row memory of the 64-channel levels in the TRAINING step): m = [xh r ; r ; 0] and its fused
    backward -- dfeat and the gradients of the LayerNorm affine, the Gram matrix, hvec and evar (summed over all pixels through
    per-workgroup partial rows) -- against fp64 autograd of the plain formula."""
    g = _g(61)
    C, Cm, N = 64, 72, H * W
    feat = torch.randn(B, C, H, W, generator=g)
    g1, b1 = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    Wm = torch.randn(256, C, generator=g) * 0.15
    bm = torch.randn(256, generator=g) * 0.1
    gram, hvec, evar = ops.memory_variance_form(Wm.to(DEV), bm.to(DEV))
    gram, hvec = gram.cpu(), hvec.cpu()
    dm = torch.randn(B, Cm, N, generator=g)
    leaves = [_leaf(feat), _leaf(g1), _leaf(b1), _leaf(gram), _leaf(hvec), _leaf(torch.tensor([[[evar]]]))]
    m = T.CompactMemFn.apply(*leaves, Cm, 1e-5, 1e-5)
    m.backward(dm.to(DEV))
    ref = [t.double().requires_grad_(True) for t in (feat, g1, b1, gram, hvec, torch.tensor([[[evar]]]))]
    x = ref[0].reshape(B, C, N)
    mu = x.mean(1, keepdim=True)
    xn = (x - mu) / torch.sqrt(((x - mu) ** 2).mean(1, keepdim=True) + 1e-5)
    xh = xn * ref[1][None, :, None] + ref[2][None, :, None]
    v = torch.einsum('bcn,cd,bdn->bn', xh, ref[3], xh) + 2 * torch.einsum('c,bcn->bn', ref[4], xh) + ref[5].reshape(())
    r = (v + 1e-5).rsqrt()
    mr = torch.cat([xh * r[:, None], r[:, None], torch.zeros(B, Cm - C - 1, N, dtype=torch.float64)], 1)
    mr.backward(dm.double())
    assert _rel(m, mr) < 3e-6
    for name, a, b_ in zip(("dfeat", "dg1", "db1", "dgram", "dhvec", "devar"), leaves, ref):
        e = _rel(a.grad, b_.grad)
        print(f"compact memory B={B} {H}x{W}: {name} rel err {e:.2e}")
        # devar = sum of dv over all pixels: a signed sum with cancellation (its relative error is that of the terms times |sum|dv|| / |sum dv|)
        assert e < (2e-3 if name == "devar" else 5e-5), (name, e)


@pytest.mark.parametrize("B,C,K,H,W", [(3, 64, 5, 32, 32), (4, 12, 5, 13, 40), (2, 64, 5, 72, 64)])
def test_select_conv_function_vs_conv_then_gather_autograd(B, C, K, H, W):
    """SelectConvFn (r05: the output layer fused with the class pick in the training step too): forward, data gradient and per-class
    weight / bias gradients against fp64 autograd of conv2d(x, w, b) followed by the per-sample channel gather; one class has no
    sample (its gradients must be exact zeros), one has two."""
    g = _g(71)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) * 0.1
    bias = torch.randn(K, generator=g) * 0.1
    idx = torch.tensor([3, 0, 3, 1][:B], dtype=torch.int32)
    up = torch.randn(B, 1, H, W, generator=g)
    xd, wd, bd = _leaf(x), _leaf(w), _leaf(bias)
    pred = T.SelectConvFn.apply(xd, wd, bd, idx.to(DEV))
    pred.backward(up.to(DEV))
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), bias.double().requires_grad_(True)
    full = F.conv2d(xr, wr, br, padding=1)
    ref = full[torch.arange(B), idx.long()][:, None]
    ref.backward(up.double())
    assert _rel(pred, ref) < 3e-6 and _rel(xd.grad, xr.grad) < 3e-6 and _rel(wd.grad, wr.grad) < 1e-5 and _rel(bd.grad, br.grad) < 1e-5
    unused = [k for k in range(K) if k not in idx.tolist()]
    assert float(wd.grad[unused].abs().max()) == 0.0 and float(bd.grad[unused].abs().max()) == 0.0


def test_stacked_token_functions_vs_torch_autograd():
    """BLinearFn / BLinear3Fn / BLayerNormFn / StackParamsFn / StackFn / UnstackFn / JoinFn / HeadFoldInLFn / HeadFoldOutLFn (r05: the
    token chains of a net's four ScoreMapModule decoders as one stack) against torch autograd of the per-level formulas in fp64."""
    g = _g(83)
    L, R, K, N = 4, 15, 40, 24
    xs = [torch.randn(R, K, generator=g) for _ in range(L)]
    ws = [torch.randn(N, K, generator=g) * 0.2 for _ in range(L)]
    bs = [torch.randn(N, generator=g) * 0.1 for _ in range(L)]
    gs_ = [torch.rand(K, generator=g) + 0.5 for _ in range(L)]
    be = [torch.randn(K, generator=g) * 0.1 for _ in range(L)]
    up = torch.randn(L, R, N, generator=g)
    xd, wd, bd, gd, bed = ([_leaf(t) for t in ts] for ts in (xs, ws, bs, gs_, be))
    W, Bb, G, Be = T.StackParamsFn.apply(L, 4, *wd, *bd, *gd, *bed)
    X = T.StackFn.apply(*xd)
    Y = T.BLinearFn.apply(T.BLayerNormFn.apply(X, G, Be, 1e-5), W, Bb)
    ys = T.UnstackFn.apply(Y)
    sum((y * up[l].to(DEV)).sum() for l, y in enumerate(ys)).backward()
    for l in range(L):
        xr, wr, br, gr, ber = (t.double().requires_grad_(True) for t in (xs[l], ws[l], bs[l], gs_[l], be[l]))
        yr = F.layer_norm(xr, (K,), gr, ber, 1e-5) @ wr.t() + br
        (yr * up[l].double()).sum().backward()
        assert _rel(ys[l], yr) < 3e-6
        for a, b_ in ((xd[l], xr), (wd[l], wr), (bd[l], br), (gd[l], gr), (bed[l], ber)):
            assert _rel(a.grad, b_.grad) < 2e-5
    # packed q | k | v projection
    w3 = [[torch.randn(N, K, generator=g) * 0.2 for _ in range(L)] for _ in range(3)]
    w3d = [[_leaf(t) for t in ws_] for ws_ in w3]
    xd2 = [_leaf(t) for t in xs]
    Wq, Wk, Wv = T.StackParamsFn.apply(L, 3, *w3d[0], *w3d[1], *w3d[2])
    up3 = torch.randn(L, R, 3 * N, generator=g)
    (T.BLinear3Fn.apply(T.StackFn.apply(*xd2), Wq, Wk, Wv) * up3.to(DEV)).sum().backward()
    for l in range(L):
        xr = xs[l].double().requires_grad_(True)
        wr = [w3[i][l].double().requires_grad_(True) for i in range(3)]
        (torch.cat([xr @ w.t() for w in wr], 1) * up3[l].double()).sum().backward()
        assert _rel(xd2[l].grad, xr.grad) < 2e-5
        for i in range(3):
            assert _rel(w3d[i][l].grad, wr[i].grad) < 2e-5
    # per-level head folds on a stacked query matrix / into a stacked output
    heads, dh, Cm = 4, 8, 12
    q = torch.randn(L, R, heads * dh, generator=g)
    wf = [torch.randn(heads * dh, Cm, generator=g) * 0.3 for _ in range(L)]
    upq = torch.randn(L, R, heads * dh, generator=g)
    qd, wfd = _leaf(q), [_leaf(t) for t in wf]
    shared, AV, parts = {}, torch.empty(L, R, heads * dh, device=DEV), []
    for l in range(L):
        y = T.HeadFoldInLFn.apply(qd, wfd[l], heads, l, shared)                     # [R, heads, Cm]
        parts.append(T.HeadFoldOutLFn.apply(y, wfd[l], heads, T._Slot(AV[l].detach())))
    (T.JoinFn.apply(T._Slot(AV), *parts) * upq.to(DEV)).sum().backward()
    qr = q.double().requires_grad_(True)
    wfr = [t.double().requires_grad_(True) for t in wf]
    tot = 0
    for l in range(L):
        wh = wfr[l].reshape(heads, dh, Cm)
        y = torch.einsum("rhd,hdn->rhn", qr[l].reshape(R, heads, dh), wh)
        tot = tot + (torch.einsum("rhn,hdn->rhd", y, wh).reshape(R, heads * dh) * upq[l].double()).sum()
    tot.backward()
    assert _rel(qd.grad, qr.grad) < 2e-5
    for l in range(L):
        assert _rel(wfd[l].grad, wfr[l].grad) < 2e-5
