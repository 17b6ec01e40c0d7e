#!/usr/bin/env python3
"""Second batch of golden vectors for the IRSDE method surface, from the REAL reference implementation
(companion of make_golden_sde.py; same import recipe, dev container only).  Covers the pieces the first file does
not hold: sde_reverse_drift / ode_reverse_drift / dispersion / forward_step / get_score_from_noise / score_fn_ /
optimal_reverse, per-sample tensor `t`, and the default scalar `mu = 0.`.  Output: tests/golden/irsde_golden2.npz (data only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_sde2.py
"""
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden_sde import InjectedNoise, load_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "irsde_golden2.npz")


def main():
    ref = load_reference()
    out = {}
    sde = ref.IRSDE(max_sigma=0.4, T=100, schedule="cosine", eps=0.01, device=torch.device("cpu"))
    g = torch.Generator().manual_seed(991)
    B, H = 4, 12
    x = torch.rand(B, 1, H, H, generator=g) * 2 - 1
    x0 = torch.rand(B, 1, H, H, generator=g) * 2 - 1
    mu = torch.rand(B, 1, H, H, generator=g) * 2 - 1
    score = torch.randn(B, 1, H, H, generator=g) * 3
    noise = torch.randn(B, 1, H, H, generator=g)
    z = torch.randn(B, 1, H, H, generator=g)
    tt = torch.tensor([3, 50, 77, 100]).reshape(B, 1, 1, 1)
    for k, v in dict(x=x, x0=x0, mu=mu, score=score, noise=noise, z=z).items():
        out[f"in/{k}"] = v.numpy()
    out["in/tt"] = tt.numpy()
    real_randn_like = torch.randn_like

    def with_noise(fn):
        torch.randn_like = InjectedNoise([z])
        try:
            return fn()
        finally:
            torch.randn_like = real_randn_like

    for tag, m in (("mu", mu), ("mu0", None)):
        if m is not None:
            sde.set_mu(m)
        else:
            sde.mu = 0.  # the constructor's default (sde_utils.py:152)
        for tname, t in (("t1", 1), ("t42", 42), ("t100", 100), ("tt", tt)):
            p = f"{tag}/{tname}"
            out[f"{p}/mu_bar"] = sde.mu_bar(x0, t).numpy()
            out[f"{p}/drift"] = sde.drift(x, t).numpy()
            out[f"{p}/sde_reverse_drift"] = sde.sde_reverse_drift(x, score, t).numpy()
            out[f"{p}/ode_reverse_drift"] = sde.ode_reverse_drift(x, score, t).numpy()
            out[f"{p}/dispersion"] = with_noise(lambda: sde.dispersion(x, t)).numpy()
            out[f"{p}/score_from_noise"] = sde.get_score_from_noise(noise, t).numpy()
            out[f"{p}/forward_step"] = with_noise(lambda: sde.forward_step(x, t)).numpy()
            out[f"{p}/reverse_sde_step_mean"] = sde.reverse_sde_step_mean(x, score, t).numpy()
            out[f"{p}/reverse_sde_step"] = with_noise(lambda: sde.reverse_sde_step(x, score, t)).numpy()
            out[f"{p}/reverse_ode_step"] = sde.reverse_ode_step(x, score, t).numpy()
            out[f"{p}/real_noise"] = sde.get_real_noise(x, x0, t).numpy()
            out[f"{p}/real_score"] = sde.get_real_score(x, x0, t).numpy()
            out[f"{p}/init_from_noise"] = sde.get_init_state_from_noise(x, noise, t).numpy()
            if not torch.is_tensor(t):
                out[f"{p}/reverse_optimum_step"] = sde.reverse_optimum_step(x, x0, t).numpy()
            out[f"{p}/weights"] = sde.weights(t).numpy()
    sde.set_mu(mu)
    out["opt/optimal_reverse_T7"] = sde.optimal_reverse(x, x0, T=7).numpy()
    sde.set_model(lambda xx, m, t, **kw: 0.8 * xx + 0.1 * m)  # an "x0-predicting" stand-in for score_fn_ (:190-194)
    out["opt/score_fn_x0pred_t9"] = sde.score_fn_(x, 9, 1.0).numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, len(out), "arrays", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
