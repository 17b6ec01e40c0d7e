"""Drivers and on-device metrics on the GPU box: image metrics vs the oracle restatement of the skimage formulas,
a bounded trainUM run (2 iterations incl. validation + checkpoint) and a testUM run that reloads it."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from instancediff_amd import ops, pipeline, testUM, trainUM  # noqa: E402
from instancediff_amd.utils.synthetic import make_batch  # noqa: E402
from oracle import metrics_ref  # noqa: E402


def test_image_metrics_vs_oracle():
    b = make_batch(3, 48, seed=9)
    pred = (b['target'] + 0.1 * b['input']).clamp(-1, 1)
    out = ops.image_metrics(pred[:, 0].cuda(), b['target'][:, 0].cuda()).cpu().numpy()
    for i in range(3):
        r, p, s = metrics_ref.metrics(pred[i, 0].numpy(), b['target'][i, 0].numpy())
        assert abs(out[i, 0] - r) < 1e-6 and abs(out[i, 1] - p) < 1e-3 and abs(out[i, 2] - s) < 2e-5, (out[i], (r, p, s))


def test_train_then_test_drivers(tmp_path):
    txt = open(pipeline.DEFAULT_YAML).read()
    txt = txt.replace("name: UM_IDDM_SM_IB", "name: drv_smoke").replace("image_size: 64", "image_size: 32")
    txt = txt.replace("T: 100", "T: 4").replace("val_freq: 3", "val_freq: 2\n  max_iters: 2").replace("save_checkpoint_freq: 8", "save_checkpoint_freq: 2")
    txt = txt.replace("path:\n", f"path:\n  root: {tmp_path}\n")
    txt = txt.replace("pth_dir: experiments/UM_IDDM_SM_IB/models", f"pth_dir: {tmp_path}/experiments/drv_smoke/models")
    txt = txt.replace("iter: latest", "iter: 2").replace("result_root: results", f"result_root: {tmp_path}/results")
    cfg = tmp_path / "cfg.yml"
    cfg.write_text(txt)
    steps = trainUM.main(["-opt", str(cfg)])
    assert steps == 2
    mdir = tmp_path / "experiments" / "drv_smoke" / "models"
    assert (mdir / "2_DN.pth").exists() and (mdir / "latest_NN.pth").exists() and (mdir / "lastest_DP_ema.pth").exists()
    assert any(f.endswith(".raw") for f in os.listdir(tmp_path / "experiments" / "drv_smoke" / "val_images"))
    res = testUM.main(["-opt", str(cfg), "--limit", "2"])
    n = sum(v['num'] for v in res.values())
    assert n == 2
    for v in res.values():
        for k in ('RMSE', 'SSIM', 'PSNR'):
            assert all(np.isfinite(x) for x in v[k])
    raws = [os.path.join(dp, f) for dp, _, fs in os.walk(tmp_path / "results") for f in fs if f.endswith(".raw")]
    assert raws and os.path.getsize(raws[0]) == 32 * 96 * 4


def test_pipeline_with_the_real_context_text_encoder():
    """The real `CLIPTextContextEncoder` (random init: no CLIP archive offline) as the nets' text_encoder argument, real-shaped class
    token ids through the tokenizer hook: a 2-step chain equals the oracle chain driven by the same encoder."""
    import torch.nn as nn
    from instancediff_amd import pipeline
    from instancediff_amd.models.drift_noise_model import create_CLIPDriftModel
    from instancediff_amd.models.SDEs import create_sde
    from instancediff_amd.models.text_encoder import CLIPTextContextEncoder
    from instancediff_amd.utils.synthetic import make_batch
    from oracle import sde_ref, unet_ref
    torch.manual_seed(5)
    enc = CLIPTextContextEncoder(context_length=42, embed_dim=512, transformer_width=512, transformer_heads=8, transformer_layers=2).eval()
    for p in enc.parameters():
        p.requires_grad_(False)
    ids = torch.randint(1, 49000, (5, 34))
    ids[:, 0] = 49406
    for k in range(5):
        ids[k, 5 + 2 * k] = 49407
        ids[k, 6 + 2 * k:] = 0
    opt = pipeline.load_options()
    train_opt = dict(opt['train'])
    train_opt['dist'] = False
    mo = opt['models']['DriftNoise']
    torch.manual_seed(0)
    model = create_CLIPDriftModel(train_opt, mo, phase="test", device=torch.device("cuda"), text_encoder=enc, class_tokens=ids)
    T = 2
    sde_opt = dict(opt['sdes'][train_opt['which_sde']])
    sde_opt['T'] = T
    sde = create_sde(model.get_nets(), sde_opt)
    sde.set_gpu(model.device)
    model.set_sde(sde)
    model.set_eval()
    batch = make_batch(2, 32, seed=9)
    g = torch.Generator().manual_seed(10)
    x_T = batch['input'] + 0.4 * torch.randn(batch['input'].shape, generator=g)
    noises = torch.randn((T,) + tuple(batch['input'].shape), generator=g)
    model.feed_data(batch)
    model.test(x_T=x_T.cuda(), noises=noises.cuda())
    out = torch.from_numpy(model.get_visuals())
    refs = []
    for key, net in (('dnet_settings', model.drift_net), ('nnet_settings', model.noise_net)):
        s = {k: v for k, v in dict(mo[key]).items() if k not in ("module_name", "class_name")}
        smm = nn.ModuleList([unet_ref.ScoreMapModule(visual_dim=mo['score_map_ngf'] * m, prompt_len=34) for m in mo['score_map_ch_mult']])
        r = unet_ref.LearnableForwardUNet_MultiScoreMap(CLIP_ScoreMapModule=smm, use_image_context=True, **s).eval()
        r.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()})
        refs.append(r)
    rsde = sde_ref.DriftSDERef(T, refs[0], refs[1], max_sigma=0.4)
    cpu_enc = CLIPTextContextEncoder(context_length=42, embed_dim=512, transformer_width=512, transformer_heads=8, transformer_layers=2).eval()
    cpu_enc.load_state_dict({k: v.cpu() for k, v in enc.state_dict().items()})
    with torch.no_grad():
        ref = rsde.reverse_ddpm(batch['input'], batch['names'], cpu_enc, x_T, noises, image_context=batch['A_emb'])
    err = float((out - ref).abs().max())
    assert torch.isfinite(out).all() and err < 5e-4, err


def test_pipeline_with_the_biomedclip_context_text_encoder():
    """CLIP_Type "BiomedCLIP" (models/drift_noise_model.py:71-77): `HFContextTextEncoder` (PubMedBERT parameter paths, 2 layers here,
    random init) as the nets' text_encoder argument -- 768-wide context tokens, BERT token ids with padding; a 2-step chain equals
    the oracle chain driven by the same encoder.  The oracle calls it with the context set expanded over the batch (as the
    reference does, drift_noise_model.py:252), the product once: both give the same text embeddings."""
    import torch.nn as nn
    from instancediff_amd import pipeline
    from instancediff_amd.models.drift_noise_model import create_CLIPDriftModel
    from instancediff_amd.models.SDEs import create_sde
    from instancediff_amd.models.text_encoder import HFContextTextEncoder
    from instancediff_amd.utils.synthetic import make_batch
    from oracle import sde_ref, unet_ref
    torch.manual_seed(6)
    enc = HFContextTextEncoder(num_hidden_layers=2).eval()
    for p in enc.parameters():
        p.requires_grad_(False)
    ids = torch.randint(1000, 30000, (5, 12))
    ids[:, 0] = 2
    for k in range(5):
        ids[k, 4 + k] = 3
        ids[k, 5 + k:] = 0
    opt = pipeline.load_options()
    train_opt = dict(opt['train'])
    train_opt['dist'] = False
    mo = dict(opt['models']['DriftNoise'])
    mo['CLIP_Type'] = "BiomedCLIP"
    torch.manual_seed(0)
    model = create_CLIPDriftModel(train_opt, mo, phase="test", device=torch.device("cuda"), text_encoder=enc, class_tokens=ids)
    assert model.token_embed_dim == 768 and tuple(model.drift_net.CLIP_ScoreMapModule[0].contexts.shape) == (1, 8, 768)
    with torch.no_grad():
        for net in (model.drift_net, model.noise_net):
            for m in net.CLIP_ScoreMapModule:
                m.contexts.mul_(5.0)  # learned context tokens of a visible size (init std 0.02)
    T = 2
    sde_opt = dict(opt['sdes'][train_opt['which_sde']])
    sde_opt['T'] = T
    sde = create_sde(model.get_nets(), sde_opt)
    sde.set_gpu(model.device)
    model.set_sde(sde)
    model.set_eval()
    batch = make_batch(2, 32, seed=9)
    g = torch.Generator().manual_seed(10)
    x_T = batch['input'] + 0.4 * torch.randn(batch['input'].shape, generator=g)
    noises = torch.randn((T,) + tuple(batch['input'].shape), generator=g)
    model.feed_data(batch)
    model.test(x_T=x_T.cuda(), noises=noises.cuda())
    out = torch.from_numpy(model.get_visuals())
    refs = []
    for key, net in (('dnet_settings', model.drift_net), ('nnet_settings', model.noise_net)):
        s = {k: v for k, v in dict(mo[key]).items() if k not in ("module_name", "class_name")}
        smm = nn.ModuleList([unet_ref.ScoreMapModule(visual_dim=mo['score_map_ngf'] * m, prompt_len=12, token_embed_dim=768) for m in mo['score_map_ch_mult']])
        r = unet_ref.LearnableForwardUNet_MultiScoreMap(CLIP_ScoreMapModule=smm, use_image_context=True, **s).eval()
        r.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()})
        refs.append(r)
    rsde = sde_ref.DriftSDERef(T, refs[0], refs[1], max_sigma=0.4)
    cpu_enc = HFContextTextEncoder(num_hidden_layers=2).eval()
    cpu_enc.load_state_dict({k: v.cpu() for k, v in enc.state_dict().items()})
    with torch.no_grad():
        ref = rsde.reverse_ddpm(batch['input'], batch['names'], cpu_enc, x_T, noises, image_context=batch['A_emb'])
    err = float((out - ref).abs().max())
    assert torch.isfinite(out).all() and err < 5e-4, err
