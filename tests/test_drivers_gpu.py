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
