"""End-to-end sampling parity (BASELINE config c1: 64x64, batch 4, 50-step reverse chain): HIP pipeline on the
GPU vs the oracle chain on the CPU with identical weights, inputs and injected noise.  Bar: |dPSNR| < 1e-3 dB
(north_star), fp32."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from instancediff_amd import pipeline  # noqa: E402
from instancediff_amd.utils.sde_utils import IRSDE  # noqa: E402
from instancediff_amd.utils.synthetic import make_batch  # noqa: E402
from oracle import sde_ref, unet_ref  # noqa: E402

DEV = "cuda"


def oracle_nets(model):
    opt = pipeline.load_options()
    mo = opt['models']['DriftNoise']
    refs = []
    for key, net in (('dnet_settings', model.drift_net), ('nnet_settings', model.noise_net)):
        s = {k: v for k, v in dict(mo[key]).items() if k not in ("module_name", "class_name")}
        smm = nn.ModuleList([unet_ref.ScoreMapModule(visual_dim=mo['score_map_ngf'] * m) for m in mo['score_map_ch_mult']])
        r = unet_ref.LearnableForwardUNet_MultiScoreMap(CLIP_ScoreMapModule=smm, use_image_context=True, **s).eval()
        r.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()})
        refs.append(r)
    return refs


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
