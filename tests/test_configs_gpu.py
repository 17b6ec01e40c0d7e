"""BASELINE.json configurations at their own sizes, against the oracle where the oracle finishes in seconds and through
size-independent properties at the full batch:

  c2 / c4  256x256, five-modality mixed batch: a B=5 slice (one image per artifact type of Configurations/config.yml:15) of the
           B=16 batch runs a 2-step injected-noise chain against the oracle's CPU chain; the full B=16 run must reproduce those five
           samples BIT FOR BIT (batch invariance: catches sample-index / stride / persistent-work-item bugs at batch 16 without
           oracle cost -- every reduction of the path is per sample, the key split of the ScoreMapModule cross-attention included).
  c5       512x512: B=1 2-step chain against the oracle, B=8 batch invariance, and the two kernels whose sizes only occur here
           (self-attention over N = 4096 tokens, ScoreMapModule cross-attention over N = 262 144 keys) against fp64 references.
  c3       training: one step at 256x256 B=2 against the oracle's torch autograd; at B=32 the gradient must equal the mean of the 16
           micro-batch gradients (linearity of the backward in the batch).
  T=1000   a full 1000-step chain (T=1000 coefficient tables, error accumulation over the whole schedule) at 32x32 B=1.

fp32 throughout; tolerance: |dPSNR| < 1e-3 dB (north_star) and max|diff| < 5e-4 against the oracle, equality for the properties.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from instancediff_amd import ops, pipeline, train_ops  # noqa: E402
from instancediff_amd.utils.synthetic import ARTIFACT_TYPES, make_batch  # noqa: E402
from oracle import sde_ref, unet_ref  # noqa: E402
from tests.test_sampling_gpu import make_scoremap_branch_visible, oracle_nets  # noqa: E402

DEV = "cuda"


def _slice(batch, n):
    return {k: (v[:n] if not isinstance(v, list) else v[:n]) for k, v in batch.items()}


def _chain(model, batch, x_T, noises):
    model.feed_data(batch)
    model.test(x_T=x_T.to(DEV), noises=noises.to(DEV))
    return torch.from_numpy(model.get_visuals()).clone()


def _oracle_chain(model, T, batch, x_T, noises):
    refs = oracle_nets(model)
    rsde = sde_ref.DriftSDERef(T, refs[0], refs[1], max_sigma=0.4)
    with torch.no_grad():
        return rsde.reverse_ddpm(batch['input'], batch['names'], unet_ref.StubTextEncoder(), x_T, noises, image_context=batch['A_emb'])


def _check_vs_oracle(out, ref, target, what):
    err = float((out - ref).abs().max())
    worst = 0.0
    for b in range(out.shape[0]):  # per image (testUM.py computes PSNR per image)
        worst = max(worst, abs(sde_ref.psnr(out[b], target[b]) - sde_ref.psnr(ref[b], target[b])))
    print(f"{what}: max|hip-oracle| {err:.3e}, worst per-image |dPSNR| {worst:.2e} dB")
    assert torch.isfinite(out).all()
    assert worst < 1e-3 and err < 5e-4, (what, worst, err)


def test_c4_c2_five_modalities_vs_oracle_and_batch16_invariance():
    T, H = 2, 256
    model, sde = pipeline.build(phase="test", device=torch.device(DEV), T=T, seed=0)
    model.set_eval()
    make_scoremap_branch_visible(model)  # gamma 0.3, decoder biases de-zeroed: memproj / Gram / folds / key split carry weight
    b16 = make_batch(16, H, seed=2024)
    assert b16['names'][:5] == ARTIFACT_TYPES  # one image per modality in the first five
    g = torch.Generator().manual_seed(2025)
    x_T = b16['input'] + 0.4 * torch.randn(b16['input'].shape, generator=g)
    noises = torch.randn((T,) + tuple(b16['input'].shape), generator=g)
    out16 = _chain(model, b16, x_T, noises)
    out5 = _chain(model, _slice(b16, 5), x_T[:5], noises[:, :5].contiguous())
    assert out16.shape == (16, 1, H, H) and torch.isfinite(out16).all()
    assert torch.equal(out16[:5], out5), f"batch-16 run differs from the batch-5 run on the same samples: {(out16[:5] - out5).abs().max()}"
    # a different neighbourhood as well: samples 8..15 alone
    b8 = {k: v[8:] for k, v in b16.items()}
    out8 = _chain(model, b8, x_T[8:], noises[:, 8:].contiguous())
    assert torch.equal(out16[8:], out8)
    ref5 = _oracle_chain(model, T, _slice(b16, 5), x_T[:5], noises[:, :5])
    _check_vs_oracle(out5, ref5, b16['target'][:5], "c4 256x256 five modalities")


def test_c5_512_chain_vs_oracle_and_batch8_invariance():
    T, H = 2, 512
    model, sde = pipeline.build(phase="test", device=torch.device(DEV), T=T, seed=0)
    model.set_eval()
    make_scoremap_branch_visible(model)  # gamma 0.3, decoder biases de-zeroed: memproj / Gram / folds / key split carry weight
    b8 = make_batch(8, H, seed=512)
    g = torch.Generator().manual_seed(513)
    x_T = b8['input'] + 0.4 * torch.randn(b8['input'].shape, generator=g)
    noises = torch.randn((T,) + tuple(b8['input'].shape), generator=g)
    out8 = _chain(model, b8, x_T, noises)
    out1 = _chain(model, _slice(b8, 1), x_T[:1], noises[:, :1].contiguous())
    assert torch.isfinite(out8).all()
    assert torch.equal(out8[:1], out1), f"{(out8[:1] - out1).abs().max()}"
    out_last = _chain(model, {k: v[7:] for k, v in b8.items()}, x_T[7:], noises[:, 7:].contiguous())
    assert torch.equal(out8[7:], out_last)
    ref1 = _oracle_chain(model, T, _slice(b8, 1), x_T[:1], noises[:, :1])
    _check_vs_oracle(out1, ref1, b8['target'][:1], "c5 512x512")
    # BASELINE c5 names "fp16 MFMA attention": the same chain with the mid self-attention (N = 4096 tokens) on the fp16 matrix cores,
    # operands clamped to +-255 -- the reference's own half-precision form (Attention_flash, models/_modified_BiomedCLIP.py:509-513)
    # -- against the SAME fp32 oracle chain and the same 1e-3 dB bar
    ops.ATTN_DTYPE = "f16"
    try:
        out1_h = _chain(model, _slice(b8, 1), x_T[:1], noises[:, :1].contiguous())
    finally:
        ops.ATTN_DTYPE = "f32"
    assert not torch.equal(out1_h, out1), "the fp16 attention kernel did not run"
    _check_vs_oracle(out1_h, ref1, b8['target'][:1], "c5 512x512, fp16 MFMA self-attention")
    print(f"c5 fp16 attention vs fp32 path: max|diff| {float((out1_h - out1).abs().max()):.3e}")


def test_c5_self_attention_4096_tokens_vs_fp64():
    """the mid-block self-attention at the 512x512 input's lowest level: N = 64*64 = 4096 tokens, 4 heads x 64"""
    g = torch.Generator().manual_seed(40)
    B, C, H, heads = 2, 256, 64, 4
    qkv = torch.randn(B, 3 * C, H, H, generator=g) * 0.7
    q, k, v = [t.reshape(B, heads, C // heads, H * H).double() for t in qkv.chunk(3, dim=1)]
    scale = (C // heads) ** -0.5
    att = (torch.einsum('bhcn,bhcm->bhnm', q, k) * scale).softmax(-1)
    ref = torch.einsum('bhnm,bhcm->bhcn', att, v).reshape(B, C, H, H)
    out = ops.attn_self(qkv.to(DEV), heads, scale)
    err = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
    print(f"attn_self N=4096: rel err {err:.2e}")
    assert err < 1e-5


@pytest.mark.parametrize("Cm", [72, 136])
def test_c5_scoremap_cross_attention_262144_keys_vs_fp64(Cm):
    """ScoreMapModule cross-attention at level 0 of a 512x512 input: 20 query rows against N = 512*512 keys"""
    g = torch.Generator().manual_seed(41)
    B, Nq, heads, N = 2, 5, 4, 512 * 512
    qf = torch.randn(B, Nq, heads, Cm, generator=g) * 0.3
    mem = torch.randn(B, Cm, N, generator=g)
    scale = 0.125
    s = torch.einsum('bqhc,bcn->bqhn', qf.double(), mem.double()) * scale
    ref = torch.einsum('bqhn,bcn->bqhc', s.softmax(-1), mem.double())
    out = ops.smm_xattn(qf.to(DEV), mem.to(DEV), scale)
    err = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
    print(f"smm_xattn N=262144 Cm={Cm}: rel err {err:.2e}")
    assert err < 2e-5
    # batch invariance of the key split: sample 1 alone gives the same bits
    out1 = ops.smm_xattn(qf[1:].contiguous().to(DEV), mem[1:].contiguous().to(DEV), scale)
    assert torch.equal(out[1:], out1)


# ---------------------------------------------------------------------------------------------------
def _set_train_inputs(model, sde, batch, t, eps):
    model.input = batch['input'].to(DEV)
    model.target = batch['target'].to(DEV)
    model.names = list(batch['names'])
    model.A_emb = batch['A_emb'].to(DEV)
    model.t, model.drift_noised_x, _, model.std_noise, _ = sde.forward_diffusion(model.target, model.input, t=t, eps=eps.to(DEV))


def _flat_grads(model):
    return torch.cat([g.reshape(-1) for g in model.drift_optimizer.flat_grads() + model.noise_optimizer.flat_grads()]).clone()


def _train_step_vs_oracle(H, W, seed, what):
    B, T_ = 2, 100
    model, sde = pipeline.build(phase="train", device=torch.device(DEV), T=T_, seed=0, score_map_dropout=0.0)
    model.set_train()
    with torch.no_grad():  # let the score-map branch carry gradient signal (gamma is 1e-4 at init)
        for net in (model.drift_net, model.noise_net):
            for m in net.CLIP_ScoreMapModule:
                m.gamma.fill_(0.3)
    rd, rn = oracle_nets(model)
    rd.train(), rn.train()
    te = unet_ref.StubTextEncoder()
    batch = make_batch(B, H, W, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    t = torch.tensor([[[[7]]], [[[63]]]])
    eps = torch.randn(batch['input'].shape, generator=g)
    osde = sde_ref.DriftSDERef(T_, rd, rn, max_sigma=0.4)
    _, x_t, _, std_noise, _ = osde.forward_diffusion(batch['target'], batch['input'], t, eps)
    tt = t.reshape(-1)
    pd, dsm = rd(x_t - batch['input'], batch['input'], tt, batch['names'], te, image_context=batch['A_emb'])
    pn, nsm = rn(x_t - batch['input'], x_t, tt, batch['names'], te, image_context=batch['A_emb'])
    tgt = batch['input'] - batch['target']

    def pyr(sms, lab):  # drift_noise_model.py:234-240 with torchvision-0.14 tensor Resize semantics (bilinear, no antialias)
        tot = 0
        for i, sm in enumerate(sms):
            lb = lab if i == 0 else F.interpolate(lab, size=(H >> i, W >> i), mode="bilinear", align_corners=False, antialias=False)
            tot = tot + F.mse_loss(sm, lb)
        return tot / 2.0
    l0 = F.mse_loss(pd, tgt) + F.mse_loss(pn, std_noise) + pyr(dsm, tgt) + pyr(nsm, std_noise)
    l0.backward()
    _set_train_inputs(model, sde, batch, t, eps)
    rec, _, _, _ = train_ops.forward_backward_inputRes(model)
    r = rec.cpu()
    loss = float(r[0] + r[1] + r[2:6].sum() / 2 + r[6:10].sum() / 2)
    assert abs(loss - float(l0.detach())) < 2e-5 * abs(float(l0.detach())), (loss, float(l0.detach()))
    worst = 0.0
    for net, ref, tag in ((model.drift_net, rd, "d"), (model.noise_net, rn, "n")):
        refg = dict(ref.named_parameters())
        for k, p in net.named_parameters():
            rg = refg[k].grad
            scale = float(rg.abs().max())
            if scale < 1e-12:
                assert float(p.grad.abs().max()) < 1e-9, (tag, k)
                continue
            e = float((p.grad.cpu() - rg).abs().max()) / scale
            worst = max(worst, e)
            assert e < 2e-3, (tag, k, e)
    print(f"{what} B=2: loss {loss:.6f} (oracle {float(l0.detach()):.6f}), worst relative parameter-gradient error {worst:.2e}")


def test_c3_train_step_256_vs_oracle_autograd():
    _train_step_vs_oracle(256, 256, 31, "c3 256x256")


def test_train_step_native_224_vs_oracle_autograd():
    """The training step at the reference's native 224 x 224 (data/MedSpeckle.py:44-45): levels 224 / 112 / 56 / 28 -- partial 16x32
    patches in the forward and data-gradient convs, the F(4x4,3x3) weight gradient on the two upper levels (224 = 14 x 16, 112 = 7 x
    16), the F(2x2,3x3) / direct forms below (56 and 28 are not multiples of 16), GroupNorm backward on ragged partial grids; loss and
    every parameter gradient against the oracle's autograd."""
    _train_step_vs_oracle(224, 224, 41, "training step 224x224")


def test_train_step_non_square_96x160_vs_oracle_autograd():
    """H != W in the backward: 96 x 160 (48x80 / 24x40 / 12x20 below)."""
    _train_step_vs_oracle(96, 160, 51, "training step 96x160")


def test_c3_batch32_gradient_is_mean_of_microbatch_gradients():
    B, H, T_, MB = 32, 256, 100, 2
    model, sde = pipeline.build(phase="train", device=torch.device(DEV), T=T_, seed=0, score_map_dropout=0.0)
    model.set_train()
    with torch.no_grad():
        for net in (model.drift_net, model.noise_net):
            for m in net.CLIP_ScoreMapModule:
                m.gamma.fill_(0.3)
    batch = make_batch(B, H, seed=33)
    g = torch.Generator().manual_seed(34)
    t = torch.randint(1, T_ + 1, (B, 1, 1, 1), generator=g)
    eps = torch.randn(batch['input'].shape, generator=g)
    _set_train_inputs(model, sde, batch, t, eps)
    rec, _, _, _ = train_ops.forward_backward_inputRes(model)
    full = _flat_grads(model)
    rec_full = rec.clone()
    acc = torch.zeros_like(full, dtype=torch.float64)
    rec_acc = torch.zeros(10, dtype=torch.float64, device=DEV)
    for i in range(0, B, MB):
        mb = {k: v[i:i + MB] for k, v in batch.items()}
        _set_train_inputs(model, sde, mb, t[i:i + MB], eps[i:i + MB])
        rec, _, _, _ = train_ops.forward_backward_inputRes(model)
        acc += _flat_grads(model).double()
        rec_acc += rec.double()
    acc /= B // MB
    rec_acc /= B // MB
    assert torch.isfinite(full).all() and float(full.abs().max()) > 0
    err = float((full.double() - acc).abs().max() / acc.abs().max())
    lerr = float((rec_full.double() - rec_acc).abs().max() / rec_acc.abs().max())
    print(f"c3 B=32 vs 16 micro-batches of 2: gradient rel err {err:.2e}, loss-record rel err {lerr:.2e}")
    assert err < 2e-5 and lerr < 1e-5
    # the full step then runs (Adam on the flat buffers) and the loss is finite
    _set_train_inputs(model, sde, batch, t, eps)
    loss, _ = model.optimize_parameters()
    assert math.isfinite(loss)


def test_full_1000_step_chain_vs_oracle():
    """T = 1000 (BASELINE configs 2, 5): the whole schedule -- 1000-row coefficient tables, graph replay across every t, error
    accumulation over 1000 dependent steps -- at a size the oracle finishes in about a minute (2000 oracle UNet forwards)."""
    T, B, H = 1000, 1, 32
    model, sde = pipeline.build(phase="test", device=torch.device(DEV), T=T, seed=0)
    model.set_eval()
    batch = make_batch(B, H, seed=1000)
    g = torch.Generator().manual_seed(1001)
    x_T = batch['input'] + 0.4 * torch.randn(batch['input'].shape, generator=g)
    noises = torch.randn((T,) + tuple(batch['input'].shape), generator=g)
    out = _chain(model, batch, x_T, noises)
    ref = _oracle_chain(model, T, batch, x_T, noises)
    _check_vs_oracle(out, ref, batch['target'], "1000-step chain 32x32")


def test_native_224_chain_vs_oracle_on_the_winograd_path():
    """The reference's native resolution (data/MedSpeckle.py:44-45, drift_noise_model.py:234): 224 = 7 x 32, lower levels 112 / 56 /
    28 are not multiples of the 8x32 / 16x32 patches.  All four levels run on the Winograd kernels with masked partial patches (224 /
    112 / 56 on the 16x32-item F(4x4,3x3) kernel, 28x28 -- one patch column wide -- on its half-patch form); chain parity against the
    oracle, B=2 batch invariance."""
    T, H = 2, 224
    model, sde = pipeline.build(phase="test", device=torch.device(DEV), T=T, seed=0)
    model.set_eval()
    make_scoremap_branch_visible(model)  # gamma 0.3, decoder biases de-zeroed: memproj / Gram / folds / key split carry weight
    b2 = make_batch(2, H, seed=224)
    g = torch.Generator().manual_seed(225)
    x_T = b2['input'] + 0.4 * torch.randn(b2['input'].shape, generator=g)
    noises = torch.randn((T,) + tuple(b2['input'].shape), generator=g)
    # which kernel serves a ResBlock conv at each level
    lib = ops._lib.load()
    algos = {}
    for size, C in ((224, 64), (112, 64), (56, 128), (28, 256)):
        x = torch.randn(1, C, size, size, device=DEV)
        w = ops.pack_conv_weight(torch.randn(C, C, 3, 3, device=DEV) * 0.02)
        ops.conv2d(x, w, None, 3, C, want_stats=True)
        algos[size] = lib.idiff_conv2d_last_algo()
    assert algos == {224: 3, 112: 3, 56: 3, 28: 4}, algos   # F(4x4,3x3): 16x32-pixel items down to 16 per sample, 8x32 items below
    out2 = _chain(model, b2, x_T, noises)
    out1 = _chain(model, {k: v[1:] for k, v in b2.items()}, x_T[1:], noises[:, 1:].contiguous())
    assert torch.equal(out2[1:], out1)
    ref = _oracle_chain(model, T, b2, x_T, noises)
    _check_vs_oracle(out2, ref, b2['target'], "224x224 (native)")


def test_non_square_160x288_chain_vs_oracle():
    """H != W: 160 x 288 (levels 160x288 / 80x144 / 40x72 / 20x36): whole 16x32 patches at the top, partial patches in x at 80x144,
    in both directions at 40x72, 720 tokens in the mid attention, N = 46 080 / 11 520 / 2 880 / 720 keys in the ScoreMapModules;
    2-step chain against the oracle, B = 2 batch invariance."""
    T, H, W = 2, 160, 288
    model, sde = pipeline.build(phase="test", device=torch.device(DEV), T=T, seed=0)
    model.set_eval()
    make_scoremap_branch_visible(model)
    b2 = make_batch(2, H, W, seed=160)
    g = torch.Generator().manual_seed(161)
    x_T = b2['input'] + 0.4 * torch.randn(b2['input'].shape, generator=g)
    noises = torch.randn((T,) + tuple(b2['input'].shape), generator=g)
    out2 = _chain(model, b2, x_T, noises)
    assert tuple(out2.shape[-2:]) == (H, W)
    out1 = _chain(model, {k: v[1:] for k, v in b2.items()}, x_T[1:], noises[:, 1:].contiguous())
    assert torch.equal(out2[1:], out1)
    ref = _oracle_chain(model, T, b2, x_T, noises)
    _check_vs_oracle(out2, ref, b2['target'], "160x288 (non-square)")


def test_single_scoremap_module_option_chain_and_gradients_vs_oracle():
    """`if_MultiScoreMap: False` (reference models/drift_noise_model.py:113-114,130-131: ONE default ScoreMapModule() handed to the net; r05):
    the module sits on the full-resolution level, its score map is embedded into `score_map_chan` channels of that level's skip and is the
    one score map returned.  A 3-step injected-noise chain and one training forward / backward (loss record, every parameter gradient)
    against the oracle built the same way; the pyramid loss has its single term."""
    import copy
    import torch.nn as nn
    import torch.nn.functional as F
    from instancediff_amd import train_ops as T_
    from oracle import sde_ref, unet_ref
    opt = copy.deepcopy(pipeline.load_options())
    mo = opt['models']['DriftNoise']
    mo['if_MultiScoreMap'] = False
    mo['dnet_settings']['if_MultiScoreMap'] = False
    mo['nnet_settings']['if_MultiScoreMap'] = False
    T, B, H = 3, 2, 64
    model, sde = pipeline.build(opt=opt, phase="train", device=torch.device(DEV), T=T, seed=0, score_map_dropout=0.0)
    assert not isinstance(model.drift_prompt, nn.ModuleList) and model.drift_net.n_sm == 1 and len(model.drift_net.sm_embed) == 1
    with torch.no_grad():
        for net in (model.drift_net, model.noise_net):
            net.CLIP_ScoreMapModule.gamma.fill_(0.3)
    refs = []
    for key, net in (('dnet_settings', model.drift_net), ('nnet_settings', model.noise_net)):
        s = {k: v for k, v in dict(mo[key]).items() if k not in ("module_name", "class_name")}
        r = unet_ref.LearnableForwardUNet_MultiScoreMap(CLIP_ScoreMapModule=unet_ref.ScoreMapModule(visual_dim=64), use_image_context=True, **s)
        r.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()})
        refs.append(r)
    batch = make_batch(B, H, seed=5, mixed=True)
    te = unet_ref.StubTextEncoder()
    # ---- training forward / backward ----
    model.set_train()
    g = torch.Generator().manual_seed(3)
    t = torch.tensor([[[[2]]], [[[3]]]])
    eps = torch.randn(batch['input'].shape, generator=g)
    model.input, model.target = batch['input'].to(DEV), batch['target'].to(DEV)
    model.names, model.A_emb = batch['names'], batch['A_emb'].to(DEV)
    model.t, model.drift_noised_x, _, model.std_noise, _ = sde.forward_diffusion(model.target, model.input, t=t, eps=eps.to(DEV))
    rec, _, _, _ = T_.forward_backward_inputRes(model)
    r_ = rec.cpu()
    assert float(r_[3:6].abs().sum()) == 0.0 and float(r_[7:10].abs().sum()) == 0.0   # one pyramid term per net
    loss = float(r_[0] + r_[1] + r_[2] / 2 + r_[6] / 2)
    osde = sde_ref.DriftSDERef(T, refs[0], refs[1], max_sigma=0.4)
    _, x_t, _, std_noise, _ = osde.forward_diffusion(batch['target'], batch['input'], t, eps)
    tt = t.reshape(-1)
    refs[0].train(), refs[1].train()
    pd, dsm = refs[0](x_t - batch['input'], batch['input'], tt, batch['names'], te, image_context=batch['A_emb'])
    pn, nsm = refs[1](x_t - batch['input'], x_t, tt, batch['names'], te, image_context=batch['A_emb'])
    assert len(dsm) == 1 and len(nsm) == 1
    tgt = batch['input'] - batch['target']
    l0 = F.mse_loss(pd, tgt) + F.mse_loss(pn, std_noise) + F.mse_loss(dsm[0], tgt) / 2 + F.mse_loss(nsm[0], std_noise) / 2
    l0.backward()
    assert abs(loss - float(l0.detach())) < 2e-5 * abs(float(l0.detach()))
    worst = 0.0
    for net, ref in ((model.drift_net, refs[0]), (model.noise_net, refs[1])):
        refg = dict(ref.named_parameters())
        for k, p_ in net.named_parameters():
            rg = refg[k].grad
            scale = float(rg.abs().max())
            if scale < 1e-12:
                continue
            worst = max(worst, float((p_.grad.cpu() - rg).abs().max()) / scale)
    assert worst < 1e-3, worst
    # ---- sampling chain ----
    model.set_eval()
    refs[0].eval(), refs[1].eval()
    gch = torch.Generator().manual_seed(6)
    x_T = batch['input'] + 0.4 * torch.randn(batch['input'].shape, generator=gch)
    noises = torch.randn((T,) + tuple(batch['input'].shape), generator=gch)
    out = _chain(model, batch, x_T, noises)
    with torch.no_grad():
        ref = osde.reverse_ddpm(batch['input'], batch['names'], te, x_T, noises, image_context=batch['A_emb'])
    err = float((out - ref).abs().max())
    print(f"if_MultiScoreMap False: loss {loss:.6f}, worst gradient error {worst:.2e}, chain max diff {err:.2e}")
    assert err < 1e-4
