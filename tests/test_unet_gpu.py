"""UNet / ScoreMapModule forward parity: HIP path (GPU) vs the oracle module (CPU fp32) on identical weights
and inputs.  fp32 everywhere; tolerance covers summation-order differences only."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from instancediff_amd.models.modules import create_net  # noqa: E402
from instancediff_amd.models.modules.MSM_degEmb_Unet import ARTIFACT_TYPES, ScoreMapModule  # noqa: E402
from instancediff_amd.models.text_encoder import StubTextEncoder  # noqa: E402
from oracle import unet_ref  # noqa: E402

SETTINGS = dict(module_name="MSM_degEmb_Unet", class_name="LearnableForwardUNet_MultiScoreMap", in_nc=2, out_nc=5, nf=64,
                ch_mult=[1, 2, 4, 4], context_dim=512, text_module="scoremap", score_map_chan=16, if_MultiScoreMap=True,
                score_map_ch_mult=[1, 1, 2, 4], score_map_ngf=16)  # Configurations/config.yml:106-118


def build_pair(seed=0, use_image_context=True, text_module="scoremap"):
    torch.manual_seed(seed)
    s = dict(SETTINGS, use_image_context=use_image_context, use_degra_context=False, text_module=text_module)
    smm = nn.ModuleList([ScoreMapModule(visual_dim=64 * m) for m in [1, 1, 2, 4]]) if text_module == "scoremap" else None
    net = create_net(s, CLIP_ScoreMapModule=smm).eval()
    # make the tiny-gamma / zero-init paths visible in the comparison
    with torch.no_grad():
        if smm is not None:
            for m in smm:
                m.gamma.fill_(0.5)
                for p in m.context_decoder.parameters():
                    if p.dim() == 1 and p.abs().sum() == 0:
                        p.normal_(0, 0.02)
    rs = {k: v for k, v in s.items() if k not in ("module_name", "class_name")}
    rsmm = nn.ModuleList([unet_ref.ScoreMapModule(visual_dim=64 * m) for m in [1, 1, 2, 4]]) if smm is not None else None
    ref = unet_ref.LearnableForwardUNet_MultiScoreMap(CLIP_ScoreMapModule=rsmm, **rs).eval()
    missing, unexpected = ref.load_state_dict(net.state_dict(), strict=True)
    assert not missing and not unexpected
    return net, ref


def inputs(B, H, M=1, seed=3):
    g = torch.Generator().manual_seed(seed)
    xa = torch.randn(B, 1, H, H, generator=g) * 0.4
    xb = torch.rand(B, 1, H, H, generator=g) * 2 - 1
    t = torch.randint(1, 100, (B,), generator=g)
    names = [ARTIFACT_TYPES[(i * 3 + 1) % 5] for i in range(B)]
    ctx = torch.nn.functional.normalize(torch.randn(B, M, 512, generator=g), dim=-1)
    return xa, xb, t, names, ctx


def rel(a, b):
    return float((a.double().cpu() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-12))


def test_state_dict_keys_match_oracle():
    net, ref = build_pair()
    assert list(net.state_dict().keys()) == list(ref.state_dict().keys())
    assert any("CLIP_ScoreMapModule" in k for k in net.state_dict())  # drift_noise_model.py:718-719


@pytest.mark.parametrize("H,B,M", [(64, 2, 1), (32, 3, 1), (64, 1, 4)])
def test_unet_forward_parity(H, B, M):
    net, ref = build_pair()
    te_ref = StubTextEncoder()
    xa, xb, t, names, ctx = inputs(B, H, M)
    with torch.no_grad():
        p_ref, sm_ref = ref(xa, xb, t, names, te_ref, image_context=ctx)
    net = net.cuda()
    te = StubTextEncoder().cuda()
    with torch.no_grad():
        p, sm = net(xa.cuda(), xb.cuda(), t.cuda(), names, te, image_context=ctx.cuda())
    assert p.shape == p_ref.shape == (B, 1, H, H)
    assert rel(p, p_ref) < 2e-4, rel(p, p_ref)
    assert len(sm) == 4
    for i, (a, b) in enumerate(zip(sm, sm_ref)):
        assert a.shape == b.shape == (B, 1, H >> i, H >> i)
        assert rel(a, b) < 2e-4, (i, rel(a, b))


def test_unet_no_scoremap_no_context_scalar_t():
    net, ref = build_pair(use_image_context=False, text_module="none")
    xa, xb, t, names, ctx = inputs(2, 32)
    with torch.no_grad():
        p_ref = ref(xa, xb, 37.0, names, None)
    net = net.cuda()
    with torch.no_grad():
        p = net(xa.cuda(), xb.cuda(), 37.0, names, None)
    assert torch.is_tensor(p) and rel(p, p_ref) < 2e-4


def test_forward_without_gpu_raises():
    net, _ = build_pair()
    xa, xb, t, names, ctx = inputs(1, 32)
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            net(xa, xb, t, names, StubTextEncoder(), image_context=ctx)
