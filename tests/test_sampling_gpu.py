"""End-to-end sampling parity (BASELINE config c1: 64x64, batch 4, 50-step reverse chain): HIP pipeline on the
GPU vs the oracle chain on the CPU with identical weights, inputs and injected noise.  Bar: |dPSNR| < 1e-3 dB
(north_star), fp32."""
import collections

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from instancediff_amd import ops, pipeline  # noqa: E402
from instancediff_amd.utils.sde_utils import IRSDE  # noqa: E402
from instancediff_amd.utils.synthetic import make_batch  # noqa: E402
from oracle import sde_ref, unet_ref  # noqa: E402

DEV = "cuda"


def oracle_nets(model, decoder_type="ContextDecoder", if_flash=False):
    opt = pipeline.load_options()
    mo = opt['models']['DriftNoise']
    refs = []
    for key, net in (('dnet_settings', model.drift_net), ('nnet_settings', model.noise_net)):
        s = {k: v for k, v in dict(mo[key]).items() if k not in ("module_name", "class_name")}
        smm = nn.ModuleList([unet_ref.ScoreMapModule(visual_dim=mo['score_map_ngf'] * m, decoder_type=decoder_type, if_flash=if_flash)
                             for m in mo['score_map_ch_mult']])
        r = unet_ref.LearnableForwardUNet_MultiScoreMap(CLIP_ScoreMapModule=smm, use_image_context=True, **s).eval()
        r.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()})
        refs.append(r)
    return refs


def make_scoremap_branch_visible(model, gamma=0.3, seed=77):
    """The ScoreMapModule's decoder contribution enters the score maps as gamma * out_proj(...) with gamma = 1e-4 at init, and the
    decoder's biases are zero-initialised: at init the memory projection, Gram-matrix variance, query / value folds, the key split
    and its combine add 1e-4 of their error to a chain's output.  Raise gamma and de-zero those vectors (both nets; the oracle copies
    the state dict afterwards) so those kernels carry weight in the compared images."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for net in (model.drift_net, model.noise_net):
            for m in net.CLIP_ScoreMapModule:
                m.gamma.fill_(gamma)
                for p in m.context_decoder.parameters():
                    if p.dim() == 1 and float(p.abs().sum()) == 0:
                        p.copy_((torch.randn(p.shape, generator=g) * 0.02).to(p.device))
    from instancediff_amd import train_ops
    train_ops.WEIGHT_EPOCH[0] += 1  # prepared-weight caches key on it


def _long_chain(T, B, H, seed, what, request=None, expect_wino4=()):
    """T-step injected-noise chain (2 UNet forwards + update per step) against the oracle's CPU chain; expect_wino4 lists
    (Cin, Cout, Hout) conv shapes that must have been served by the F(4x4,3x3) kernels (idiff_conv2d_last_algo() 3 = 16x32-pixel
    items, 4 = half-patch form)."""
    model, sde = pipeline.build(phase="test", device=torch.device(DEV), T=T, seed=0)
    model.set_eval()
    make_scoremap_branch_visible(model)
    batch = make_batch(B, H, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    x_T = batch['input'] + 0.4 * torch.randn(batch['input'].shape, generator=g)
    noises = torch.randn((T,) + tuple(batch['input'].shape), generator=g)
    model.feed_data(batch)
    ops.ALGO_TRACE = collections.Counter()
    try:
        if request is None:
            model.test(x_T=x_T.to(DEV), noises=noises.to(DEV))
        else:
            with ops.request_conv3x3_algo(request):
                model.test(x_T=x_T.to(DEV), noises=noises.to(DEV))
        trace = ops.ALGO_TRACE
    finally:
        ops.ALGO_TRACE = None
    for cin, cout, hout in expect_wino4:
        served = {k[0] for k in trace if k[1] == 3 and k[2] == cin and k[3] == cout and k[4] == hout}
        assert served and served <= {3, 4}, f"{what}: conv {cin}->{cout} at {hout} ran on algos {served}, expected the F(4x4,3x3) kernels only"
    n4 = sum(v for k, v in trace.items() if k[0] in (3, 4))
    n2 = sum(v for k, v in trace.items() if k[0] == 1)
    out = torch.from_numpy(model.get_visuals())
    assert out.shape == (B, 1, H, H) and torch.isfinite(out).all()
    refs = oracle_nets(model)
    rsde = sde_ref.DriftSDERef(T, refs[0], refs[1], max_sigma=0.4)
    with torch.no_grad():
        ref = rsde.reverse_ddpm(batch['input'], batch['names'], unet_ref.StubTextEncoder(), x_T, noises, image_context=batch['A_emb'])
    err = float((out - ref).abs().max())
    worst = max(abs(sde_ref.psnr(out[b], batch['target'][b]) - sde_ref.psnr(ref[b], batch['target'][b])) for b in range(B))
    print(f"{what}: {T} steps, F(4x4,3x3) / F(2x2,3x3) conv calls per traced step {n4} / {n2}; max|hip-oracle| {err:.3e}, "
          f"worst per-image |dPSNR| {worst:.2e} dB")
    assert worst < 1e-3 and err < 5e-4, (what, worst, err)


def test_c1_chain_forced_through_the_f4x4_kernel():
    """BASELINE c1 (64x64, batch 4, 50 steps) once more with every 3x3 conv asking for the F(4x4,3x3) kernel: the 64x64 and 32x32
    levels (8 and fewer items per sample: F(2x2,3x3) by the library's own choice) then run on it, so the kernel that carries most of
    a 256x256 step takes part in a 50-step dependent chain.  Same bound as the c1 chain."""
    _long_chain(50, 4, 64, 1234, "c1 forced F(4x4,3x3)", request=ops.CONV_ALGO_WINOGRAD4,
                expect_wino4=[(64, 64, 64), (144, 64, 64), (64, 64, 32), (208, 128, 32)])


def test_128_chain_50_steps_through_the_f4x4_kernel_by_default_choice():
    """128x128 batch 1: levels 0 and 1 (32 and 16 items per sample) select the F(4x4,3x3) kernel on their own, 50 dependent steps."""
    _long_chain(50, 1, 128, 128, "128x128 50-step chain", expect_wino4=[(64, 64, 128), (144, 64, 128), (208, 128, 64), (128, 64, 128)])


def test_c2_shape_25_step_chain_through_the_f4x4_kernel():
    """The headline shape (256x256): 25 dependent steps, three levels on the F(4x4,3x3) kernel, ScoreMapModule memory at N = 65 536
    with the key split and the compact memory carrying weight (gamma = 0.3)."""
    _long_chain(25, 1, 256, 256, "256x256 25-step chain", expect_wino4=[(64, 64, 256), (64, 64, 128), (128, 128, 64), (144, 64, 256)])


def test_c1_drift_chain_psnr_parity():
    T, B, H = 50, 4, 64
    model, sde = pipeline.build(phase="test", device=torch.device(DEV), T=T, seed=0)
    model.set_eval()
    batch = make_batch(B, H, seed=1234)
    g = torch.Generator().manual_seed(4321)
    x_T = batch['input'] + 0.4 * torch.randn(batch['input'].shape, generator=g)
    noises = torch.randn((T,) + tuple(batch['input'].shape), generator=g)
    model.feed_data(batch)
    model.test(x_T=x_T.to(DEV), noises=noises.to(DEV))
    out = torch.from_numpy(model.get_visuals())
    assert out.shape == (B, 1, H, H) and torch.isfinite(out).all()
    refs = oracle_nets(model)
    rsde = sde_ref.DriftSDERef(T, refs[0], refs[1], max_sigma=0.4)
    with torch.no_grad():
        ref = rsde.reverse_ddpm(batch['input'], batch['names'], unet_ref.StubTextEncoder(), x_T, noises, image_context=batch['A_emb'])
    p_hip, p_ref = sde_ref.psnr(out, batch['target']), sde_ref.psnr(ref, batch['target'])
    err = float((out - ref).abs().max())
    print(f"c1: PSNR hip {p_hip:.6f} dB, oracle {p_ref:.6f} dB, max|diff| {err:.3e}")
    assert abs(p_hip - p_ref) < 1e-3
    assert err < 5e-4
    for b in range(B):  # per image as well (testUM.py computes PSNR per image)
        assert abs(sde_ref.psnr(out[b], batch['target'][b]) - sde_ref.psnr(ref[b], batch['target'][b])) < 1e-3


def test_c2_shape_short_chain_psnr_parity():
    """The headline shape itself (256x256: Winograd convs at all four levels, compact ScoreMapModule memory at two, flattened
    1x1 convs, fused output layer, two streams) on one image and a 3-step chain -- what the oracle finishes in seconds."""
    T, B, H = 3, 1, 256
    model, sde = pipeline.build(phase="test", device=torch.device(DEV), T=T, seed=0)
    model.set_eval()
    batch = make_batch(B, H, seed=77)
    g = torch.Generator().manual_seed(78)
    x_T = batch['input'] + 0.4 * torch.randn(batch['input'].shape, generator=g)
    noises = torch.randn((T,) + tuple(batch['input'].shape), generator=g)
    model.feed_data(batch)
    model.test(x_T=x_T.to(DEV), noises=noises.to(DEV))
    out = torch.from_numpy(model.get_visuals())
    refs = oracle_nets(model)
    rsde = sde_ref.DriftSDERef(T, refs[0], refs[1], max_sigma=0.4)
    with torch.no_grad():
        ref = rsde.reverse_ddpm(batch['input'], batch['names'], unet_ref.StubTextEncoder(), x_T, noises, image_context=batch['A_emb'])
    p_hip, p_ref = sde_ref.psnr(out, batch['target']), sde_ref.psnr(ref, batch['target'])
    err = float((out - ref).abs().max())
    print(f"c2 shape: PSNR hip {p_hip:.6f} dB, oracle {p_ref:.6f} dB, max|diff| {err:.3e}")
    assert abs(p_hip - p_ref) < 1e-3
    assert err < 5e-4


def test_irsde_single_network_mode_parity():
    """utils/sde_utils.py:244-261 loop with the noise network as `model(x, mu, t, **kw)` (t arrives as a python
    float = t*sample_scale), sample_T < T fast sampling, injected noise."""
    T, sT, B, H = 100, 10, 2, 32
    model, _ = pipeline.build(phase="test", device=torch.device(DEV), T=T, seed=1)
    net = model.noise_net.eval()
    batch = make_batch(B, H, seed=5)
    g = torch.Generator().manual_seed(99)
    noises = torch.randn((sT,) + tuple(batch['input'].shape), generator=g)
    eps0 = torch.randn(batch['input'].shape, generator=g)
    kw = dict(names=batch['names'], text_encoder=model.text_encoder, image_context=batch['A_emb'].to(DEV))
    sde = IRSDE(0.4, T=T, sample_T=sT, schedule='cosine', eps=0.01, device=torch.device(DEV))
    sde.set_mu(batch['input'].to(DEV))
    sde.set_model(net)
    x_T = sde.noise_state(batch['input'].to(DEV), eps=eps0.to(DEV))
    out_sde = sde.reverse_sde(x_T, noises=noises.to(DEV), **kw).cpu()
    out_ode = sde.reverse_ode(x_T, **kw).cpu()
    rnet = oracle_nets(model)[1]
    ref = sde_ref.IRSDERef(0.4, T=T, sample_T=sT, schedule='cosine', eps=0.01)
    ref.set_mu(batch['input'])
    ref.set_model(lambda x, mu, t, **k: rnet(x, mu, t, **k)[0])
    rkw = dict(names=batch['names'], text_encoder=unet_ref.StubTextEncoder(), image_context=batch['A_emb'])
    rx_T = ref.noise_state(batch['input'], eps0)
    assert torch.equal(x_T.cpu(), rx_T)
    with torch.no_grad():
        r_sde = ref.reverse_sde(rx_T, noises, **rkw)
        r_ode = ref.reverse_ode(rx_T, **rkw)
    for a, b, nm in ((out_sde, r_sde, "sde"), (out_ode, r_ode, "ode")):
        d = abs(sde_ref.psnr(a, batch['target']) - sde_ref.psnr(b, batch['target']))
        print(f"irsde {nm}: |dPSNR| {d:.2e} max|diff| {float((a - b).abs().max()):.3e}")
        assert d < 1e-3 and float((a - b).abs().max()) < 5e-4


def test_on_device_noise_is_reproducible_and_changes_with_seed():
    T, B, H = 6, 2, 32
    model, sde = pipeline.build(phase="test", device=torch.device(DEV), T=T, seed=0)
    model.set_eval()
    batch = make_batch(B, H, seed=2)
    outs = []
    for seed in (7, 7, 8):
        sde.set_seed(seed)
        model.feed_data(batch)
        sde.set_seed(seed)
        model.test()
        outs.append(torch.from_numpy(model.get_visuals()).clone())
    assert torch.equal(outs[0], outs[1])
    assert not torch.equal(outs[0], outs[2])
    assert torch.isfinite(outs[2]).all()


def test_graph_replay_is_bit_identical_to_eager_steps():
    """The captured HIP graph replays exactly the launches of the eager loop, on-device Philox noise included."""
    T, B, H = 6, 2, 32
    model, sde = pipeline.build(phase="test", device=torch.device(DEV), T=T, seed=3)
    model.set_eval()
    batch = make_batch(B, H, seed=11)
    outs = []
    for use_graph in (True, False):
        sde.hip_graph = use_graph
        sde.set_seed(99)
        model.feed_data(batch)
        model.test()
        outs.append(torch.from_numpy(model.get_visuals()).clone())
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1])


def test_failed_graph_capture_keeps_the_step_count(monkeypatch):
    """A capture that raises after the eager warm step must not cost a step: the chain equals the IDIFF_HIP_GRAPH=0 chain bit
    for bit (the warm step is a real denoising step and is counted whether or not the capture then succeeds)."""
    T, B, H = 6, 2, 32
    model, sde = pipeline.build(phase="test", device=torch.device(DEV), T=T, seed=3)
    model.set_eval()
    batch = make_batch(B, H, seed=12)
    g = torch.Generator().manual_seed(13)
    x_T = (batch['input'] + 0.4 * torch.randn(batch['input'].shape, generator=g)).to(DEV)
    noises = torch.randn((T,) + tuple(batch['input'].shape), generator=g).to(DEV)
    outs = {}
    for mode in ("eager", "graph", "broken"):
        sde.hip_graph = mode != "eager"
        if mode == "broken":
            def boom(*a, **k):
                raise RuntimeError("forced capture failure")
            monkeypatch.setattr(torch.cuda, "graph", boom)
        sde.set_seed(5)
        model.feed_data(batch)
        sde.set_seed(5)
        calls0 = sde._calls
        model.test(x_T=x_T, noises=noises)
        assert sde._calls - calls0 == T, (mode, sde._calls - calls0)
        outs[mode] = torch.from_numpy(model.get_visuals()).clone()
    assert torch.isfinite(outs["eager"]).all()
    assert torch.equal(outs["eager"], outs["graph"])
    assert torch.equal(outs["eager"], outs["broken"])


def test_philox_draws_of_mixed_sizes_never_share_counters():
    """Successive draws of one stream use disjoint Philox counter ranges whatever their sizes (training crops, then a validation
    image of another size): the second draw continues where the first ended, for IRSDE and for driftSDE incl. its graph-replayed
    steps."""
    sde = IRSDE(0.4, T=10, schedule='cosine', eps=0.01, device=torch.device(DEV))
    sde.set_seed(5)
    a = sde._randn_like(torch.empty(2, 1, 32, 32, device=DEV))
    b = sde._randn_like(torch.empty(1, 1, 16, 20, device=DEV))
    c = sde.noise_state(torch.zeros(1, 1, 8, 8, device=DEV))
    ref = ops.randn((2048 + 320 + 64,), DEV, 5, 0)
    assert torch.equal(a.reshape(-1), ref[:2048]) and torch.equal(b.reshape(-1), ref[2048:2368])
    assert torch.equal(c.reshape(-1), ref[2368:] * sde.max_sigma)
    # driftSDE: forward_diffusion draw (big), then a 4-step reverse chain on a smaller image: its x_T draw and its per-step draws
    T, H = 4, 32
    model, dsde = pipeline.build(phase="test", device=torch.device(DEV), T=T, seed=0)
    model.set_eval()
    dsde.set_seed(9)
    big = make_batch(2, 64, seed=3)
    dsde.forward_diffusion(big['target'].to(DEV), big['input'].to(DEV))         # counters [0, 2048)
    small = make_batch(1, H, seed=4)
    model.feed_data(small)                                                       # forward_diffusion again: [2048, 2304)
    off0 = dsde._off
    assert off0 == 2048 + 256
    model.test()                                                                 # x_T draw + T step draws of 256 counters each
    assert dsde._off == off0 + 256 * (1 + T)
    out = torch.from_numpy(model.get_visuals()).clone()
    # the same chain with the draws injected from the counter ranges they must have used
    stream = ops.randn((4 * (off0 + 256 * (1 + T)),), DEV, 9, 0)
    x_T = ops.axpby(small['input'].to(DEV), stream[4 * off0:4 * (off0 + 256)].reshape(1, 1, H, H).contiguous(), 1.0, dsde.max_sigma)  # reverse_ddpm's own formula
    noises = stream[4 * (off0 + 256):].reshape(T, 1, 1, H, H).contiguous()
    model.test(x_T=x_T, noises=noises)
    assert torch.equal(torch.from_numpy(model.get_visuals()), out)


def test_reverse_type_other_than_std_is_refused():
    model, sde = pipeline.build(phase="test", device=torch.device(DEV), T=4, seed=0)
    b = make_batch(1, 32, seed=1)
    with pytest.raises(NotImplementedError):
        sde.reverse_ddpm(b['input'].to(DEV), b['names'], model.text_encoder, reverse_type="scaled", image_context=b['A_emb'].to(DEV))


def test_chain_with_the_hierarchical_scaled_decoder_option():
    """model option score_map_decoder: ContextDecoder_Hierachical (TransformerDecoderLayer_scaled blocks, reference
    models/_modified_BiomedCLIP.py:552-590,1247-1308) -- a 10-step 64x64 chain against the oracle built with the same option; the
    decoder made visible (gamma 0.3) and the branch gains moved off their 0.1 init."""
    T, B, H = 10, 2, 64
    model, sde = pipeline.build(phase="test", device=torch.device(DEV), T=T, seed=0, score_map_decoder="ContextDecoder_Hierachical")
    model.set_eval()
    g = torch.Generator().manual_seed(99)
    with torch.no_grad():
        for net in (model.drift_net, model.noise_net):
            for m in net.CLIP_ScoreMapModule:
                assert type(m.context_decoder).__name__ == "ContextDecoder_Hierachical"
                for l in m.context_decoder.decoder:
                    for name in ("gamma_sa", "gamma_ca", "gamma_mlp"):
                        getattr(l, name).copy_((0.1 + 0.3 * torch.randn((1, 1, 256), generator=g)).to(DEV))
    make_scoremap_branch_visible(model)
    batch = make_batch(B, H, seed=5)
    x_T = batch['input'] + 0.4 * torch.randn(batch['input'].shape, generator=g)
    noises = torch.randn((T,) + tuple(batch['input'].shape), generator=g)
    model.feed_data(batch)
    model.test(x_T=x_T.to(DEV), noises=noises.to(DEV))
    out = torch.from_numpy(model.get_visuals())
    refs = oracle_nets(model, decoder_type="ContextDecoder_Hierachical")
    assert any("gamma_ca" in k for k in refs[0].state_dict())
    rsde = sde_ref.DriftSDERef(T, refs[0], refs[1], max_sigma=0.4)
    with torch.no_grad():
        ref = rsde.reverse_ddpm(batch['input'], batch['names'], unet_ref.StubTextEncoder(), x_T, noises, image_context=batch['A_emb'])
        # the gains matter: the plain-decoder oracle with the shared weights gives a visibly different chain
        plain = oracle_nets_drop_gains(model)
        other = sde_ref.DriftSDERef(T, plain[0], plain[1], max_sigma=0.4).reverse_ddpm(
            batch['input'], batch['names'], unet_ref.StubTextEncoder(), x_T, noises, image_context=batch['A_emb'])
    err = float((out - ref).abs().max())
    worst = max(abs(sde_ref.psnr(out[b], batch['target'][b]) - sde_ref.psnr(ref[b], batch['target'][b])) for b in range(B))
    print(f"hierarchical decoder chain: max|hip-oracle| {err:.3e}, worst |dPSNR| {worst:.2e} dB; plain-decoder oracle differs by "
          f"{float((other - ref).abs().max()):.3e}")
    assert worst < 1e-3 and err < 5e-4
    assert float((other - ref).abs().max()) > 20 * max(err, 1e-7)


def oracle_nets_drop_gains(model):
    """plain-ContextDecoder oracle nets loaded with the product's weights minus the branch gains (control for the test above)"""
    opt = pipeline.load_options()
    mo = opt['models']['DriftNoise']
    refs = []
    for key, net in (('dnet_settings', model.drift_net), ('nnet_settings', model.noise_net)):
        s = {k: v for k, v in dict(mo[key]).items() if k not in ("module_name", "class_name")}
        smm = nn.ModuleList([unet_ref.ScoreMapModule(visual_dim=mo['score_map_ngf'] * m) for m in mo['score_map_ch_mult']])
        r = unet_ref.LearnableForwardUNet_MultiScoreMap(CLIP_ScoreMapModule=smm, use_image_context=True, **s).eval()
        r.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items() if ".gamma_sa" not in k and ".gamma_ca" not in k and ".gamma_mlp" not in k})
        refs.append(r)
    return refs


# ---- the reference's half-precision form of the decoder attentions (TransformerDecoderLayer_scaled(if_flash=True) -> Attention_flash,
# models/_modified_BiomedCLIP.py:481-517,552-590): model option score_map_if_flash, a labelled reduced-precision VARIANT --------------------
def test_flash_form_token_attention_vs_emulation():
    """idiff_attn_tokens_f16_fwd against oracle/unet_ref.attention_core_flash (clamp +-255, fp16 operands / probabilities / result,
    fp32 statistics); operands beyond +-255 included so the clamp matters"""
    g = torch.Generator().manual_seed(610)
    B, K, C, heads = 3, 5, 256, 4
    qkv = torch.randn(B, K, 3 * C, generator=g) * 1.5
    qkv[0, 1, 7] = 300.0
    qkv[1, 2, C + 9] = -400.0
    qkv[2, 0, 2 * C + 11] = 290.0
    scale = (C // heads) ** -0.5
    q, k, v = qkv.split(C, dim=2)
    ref = unet_ref.attention_core_flash(q, k, v, heads, scale)
    out = ops.attn_tokens_packed_f16(qkv.to(DEV).contiguous(), heads, scale).cpu()
    plain = unet_ref.attention_core(q, k, v, heads, scale)
    err = float((out - ref).abs().max() / ref.abs().max())
    print(f"flash-form token attention: rel err vs the emulation {err:.2e}; the fp32 attention differs by {float((plain - ref).abs().max() / ref.abs().max()):.2e}")
    assert err < 2e-3  # one fp16 ulp of the result (the kernel's sum order differs from the emulation's)
    assert float(out.abs().max()) <= 255.0


@pytest.mark.parametrize("N", [1024, 4096 + 36])
def test_flash_form_cross_attention_vs_emulation(N):
    """idiff_smm_xattn_kv_f16_fwd (unfolded k / v, wave = head, key splits + merge) against the same emulation; a key count that is not a
    multiple of the 32-key block in the second case"""
    g = torch.Generator().manual_seed(611 + N)
    B, K, C, heads = 2, 5, 256, 4
    q = torch.randn(B, K, C, generator=g) * 2.0
    k = torch.randn(B, N, C, generator=g) * 1.2
    v = torch.randn(B, N, C, generator=g) * 3.0
    k[0, 3, 5], v[1, 7, 200], q[1, 2, 100] = 700.0, -900.0, 400.0
    scale = (C // heads) ** -0.5
    ref = unet_ref.attention_core_flash(q, k, v, heads, scale)
    out = ops.smm_xattn_kv_f16(q.to(DEV), k.permute(0, 2, 1).contiguous().to(DEV), v.permute(0, 2, 1).contiguous().to(DEV), heads, scale).cpu()
    out2 = ops.smm_xattn_kv_f16(q.to(DEV), k.permute(0, 2, 1).contiguous().to(DEV), v.permute(0, 2, 1).contiguous().to(DEV), heads, scale).cpu()
    assert torch.equal(out, out2)
    plain = unet_ref.attention_core(q, k, v, heads, scale)
    err = float((out - ref).abs().max() / ref.abs().max())
    print(f"flash-form cross attention N={N}: rel err vs the emulation {err:.2e}; the fp32 attention differs by {float((plain - ref).abs().max() / ref.abs().max()):.2e}")
    assert err < 2e-3
    # batch invariance: a sample alone gives the same bits (the key split is a function of N alone)
    one = ops.smm_xattn_kv_f16(q[1:].to(DEV), k[1:].permute(0, 2, 1).contiguous().to(DEV), v[1:].permute(0, 2, 1).contiguous().to(DEV), heads, scale).cpu()
    assert torch.equal(one, out[1:])


def test_flash_form_decoder_chain_vs_oracle_and_refused_in_training():
    """model options score_map_decoder: ContextDecoder_Hierachical + score_map_if_flash: a 6-step 64x64 chain against the oracle built
    with the same options (Attention_flash emulated on the CPU), against the fp32 form of the same weights (the variant must differ from
    it, by little), and the training step's loud refusal"""
    T, B, H = 6, 2, 64
    model, sde = pipeline.build(phase="test", device=torch.device(DEV), T=T, seed=0, score_map_decoder="ContextDecoder_Hierachical",
                                score_map_if_flash=True)
    model.set_eval()
    g = torch.Generator().manual_seed(199)
    with torch.no_grad():
        for net in (model.drift_net, model.noise_net):
            for m in net.CLIP_ScoreMapModule:
                assert m.context_decoder.if_flash
                for l in m.context_decoder.decoder:
                    for name in ("gamma_sa", "gamma_ca", "gamma_mlp"):
                        getattr(l, name).copy_((0.3 + 0.3 * torch.randn((1, 1, 256), generator=g)).to(DEV))
    make_scoremap_branch_visible(model)
    batch = make_batch(B, H, seed=6)
    x_T = batch['input'] + 0.4 * torch.randn(batch['input'].shape, generator=g)
    noises = torch.randn((T,) + tuple(batch['input'].shape), generator=g)

    def run():
        model.feed_data(batch)
        model.test(x_T=x_T.to(DEV), noises=noises.to(DEV))
        return torch.from_numpy(model.get_visuals())
    out = run()
    for net in (model.drift_net, model.noise_net):
        for m in net.CLIP_ScoreMapModule:
            m.context_decoder.if_flash = False
    out32 = run()
    for net in (model.drift_net, model.noise_net):
        for m in net.CLIP_ScoreMapModule:
            m.context_decoder.if_flash = True
    refs = oracle_nets(model, decoder_type="ContextDecoder_Hierachical", if_flash=True)
    assert all(l.cross_attn.flash and l.self_attn.flash for r in refs for m in r.CLIP_ScoreMapModule for l in m.context_decoder.decoder)
    with torch.no_grad():
        ref = sde_ref.DriftSDERef(T, refs[0], refs[1], max_sigma=0.4).reverse_ddpm(batch['input'], batch['names'], unet_ref.StubTextEncoder(), x_T, noises,
                                                                                     image_context=batch['A_emb'])
    err = float((out - ref).abs().max())
    worst = max(abs(sde_ref.psnr(out[b], batch['target'][b]) - sde_ref.psnr(ref[b], batch['target'][b])) for b in range(B))
    d32 = float((out - out32).abs().max())
    print(f"flash-form decoder chain: max|hip-oracle(flash)| {err:.3e}, worst |dPSNR| {worst:.2e} dB; against the fp32 form of the same weights {d32:.3e}")
    assert d32 > 0.0, "the half-precision kernels did not run"
    assert worst < 1e-3 and err < 5e-4 and d32 < 5e-3
    # training: refused loudly
    tm, _ = pipeline.build(phase="train", device=torch.device(DEV), T=T, seed=0, score_map_decoder="ContextDecoder_Hierachical", score_map_if_flash=True)
    tm.set_train()
    tm.feed_data(make_batch(2, 32, seed=3))
    with pytest.raises(RuntimeError, match="inference-only"):
        tm.optimize_parameters()
