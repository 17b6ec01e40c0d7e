#!/usr/bin/env python3
"""Generate golden vectors for the IRSDE path from the REAL reference implementation.

Runs only in the dev container (needs /root/reference).  It imports
/root/reference/utils/sde_utils.py *by path* (the reference's `utils/__init__.py` pulls in cv2,
which is absent) after inserting a stub `torchvision.utils` module (only `save_image`, used in
debug branches, is referenced; sde_utils.py:5).  Nothing from the reference is copied: the output
is data only (schedule tables, trajectories, sampled states) written to tests/golden/irsde_golden.npz.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_sde.py
"""
import importlib.util
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch

REF = "/root/reference/utils/sde_utils.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "irsde_golden.npz")


def load_reference():
    tv = types.ModuleType("torchvision")
    tvu = types.ModuleType("torchvision.utils")
    tvu.save_image = lambda *a, **k: None
    tv.utils = tvu
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.utils", tvu)
    spec = importlib.util.spec_from_file_location("ref_sde_utils", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def analytic_model(x, mu, t, **kw):
    # deterministic stand-in for the noise network; t arrives as a python float (t * sample_scale)
    # transcendental-free so the fixture is portable across host CPUs (vectorised tanh differs by an ulp)
    return 0.3 * x - 0.2 * mu + (0.01 * float(t)) * (x * mu)


class InjectedNoise:
    """Replaces torch.randn_like inside the reference call so the same noise can be replayed."""

    def __init__(self, noises):
        self.noises = list(noises)
        self.i = 0

    def __call__(self, x, *a, **k):
        n = self.noises[self.i]
        self.i += 1
        assert n.shape == x.shape
        return n.clone()


def main():
    ref = load_reference()
    out = {}
    cfgs = {
        "cos100": dict(max_sigma=0.4, T=100, schedule="cosine", eps=0.01),
        "cos100_s50": dict(max_sigma=0.4, T=100, sample_T=50, schedule="cosine", eps=0.01),
        "cos1000": dict(max_sigma=0.4, T=1000, schedule="cosine", eps=0.01),
        "lin100": dict(max_sigma=0.4, T=100, schedule="linear", eps=0.01),
        "const100": dict(max_sigma=0.4, T=100, schedule="constant", eps=0.01),
        "cos100_ms50": dict(max_sigma=50, T=100, schedule="cosine", eps=0.01),
        "cos100_eps005": dict(max_sigma=0.25, T=100, schedule="cosine", eps=0.005),
    }
    for name, kw in cfgs.items():
        sde = ref.IRSDE(device=torch.device("cpu"), **kw)
        out[f"{name}/thetas"] = sde.thetas.numpy()
        out[f"{name}/sigmas"] = sde.sigmas.numpy()
        out[f"{name}/thetas_cumsum"] = sde.thetas_cumsum.numpy()
        out[f"{name}/sigma_bars"] = sde.sigma_bars.numpy()
        out[f"{name}/dt"] = np.array(float(sde.dt), dtype=np.float64)
        out[f"{name}/dt_f32"] = sde.dt.numpy() if torch.is_tensor(sde.dt) else np.array(sde.dt)
        out[f"{name}/max_sigma"] = np.array(sde.max_sigma, dtype=np.float64)
        out[f"{name}/sample_scale"] = np.array(sde.sample_scale, dtype=np.float64)

    # ---- trajectories: reverse_sde / reverse_ode / mean-step with injected noise ------------------
    real_randn_like = torch.randn_like
    for tag, (B, H), cfgname, nsteps in [
        ("t8", (2, 8), "cos100", 3),
        ("t64", (2, 64), "cos100_s50", 3),
        ("t8full", (1, 8), "cos100_s50", 50),
    ]:
        g = torch.Generator().manual_seed(4321)
        mu = torch.rand(B, 1, H, H, generator=g) * 2 - 1
        xT = mu + 0.4 * torch.randn(B, 1, H, H, generator=g)
        noises = [torch.randn(B, 1, H, H, generator=g) for _ in range(nsteps)]
        sde = ref.IRSDE(device=torch.device("cpu"), **cfgs[cfgname])
        sde.set_mu(mu)
        sde.set_model(analytic_model)
        # reference reverse_sde always starts at t=T_arg and runs down to 1: use T=nsteps so the
        # loop covers t = nsteps..1 (table rows 1..nsteps); plus a manual high-t run below.
        torch.randn_like = InjectedNoise(noises)
        try:
            x_sde = sde.reverse_sde(xT, T=nsteps)
        finally:
            torch.randn_like = real_randn_like
        x_ode = sde.reverse_ode(xT, T=nsteps)
        out[f"{tag}/mu"] = mu.numpy()
        out[f"{tag}/xT"] = xT.numpy()
        out[f"{tag}/noises"] = torch.stack(noises).numpy()
        out[f"{tag}/x_sde"] = x_sde.numpy()
        out[f"{tag}/x_ode"] = x_ode.numpy()
        # per-step pieces at the top of the schedule (t = sample_T, sample_T-1, sample_T-2)
        x = xT.clone()
        steps = []
        ts = list(range(sde.sample_T, sde.sample_T - 3, -1))
        torch.randn_like = InjectedNoise(noises[:3])
        try:
            for t in ts:
                score = sde.score_fn(x, t, sde.sample_scale)
                xm = sde.reverse_sde_step_mean(x, score, t)
                xo = sde.reverse_ode_step(x, score, t)
                x = sde.reverse_sde_step(x, score, t)
                steps.append(torch.stack([score, xm, xo, x]))
        finally:
            torch.randn_like = real_randn_like
        out[f"{tag}/top_ts"] = np.array(ts)
        out[f"{tag}/top_steps"] = torch.stack(steps).numpy()

    # ---- training-state sampler / closed forms ----------------------------------------------------
    sde = ref.IRSDE(device=torch.device("cpu"), **cfgs["cos100"])
    g = torch.Generator().manual_seed(77)
    x0 = torch.rand(4, 1, 8, 8, generator=g) * 2 - 1
    mu = torch.rand(4, 1, 8, 8, generator=g) * 2 - 1
    torch.manual_seed(2)
    t, states = sde.generate_random_states(x0, mu)
    torch.manual_seed(2)
    t_again = torch.randint(1, 101, (4, 1, 1, 1)).long()
    eps = torch.randn_like(x0)
    out["grs/x0"] = x0.numpy()
    out["grs/mu"] = mu.numpy()
    out["grs/t"] = t.numpy()
    out["grs/eps"] = eps.numpy()
    out["grs/states"] = states.numpy()
    assert torch.equal(t, t_again)
    tt = torch.tensor([1, 37, 99, 100]).reshape(4, 1, 1, 1)
    out["cf/t"] = tt.numpy()
    out["cf/mu_bar"] = sde.mu_bar(x0, tt).numpy()
    out["cf/real_noise"] = sde.get_real_noise(states, x0, tt).numpy()
    out["cf/real_score"] = sde.get_real_score(states, x0, tt).numpy()
    out["cf/init_from_noise"] = sde.get_init_state_from_noise(states, eps, tt).numpy()
    out["cf/optimum_t37"] = sde.reverse_optimum_step(states, x0, 37).numpy()
    out["cf/optimum_t100"] = sde.reverse_optimum_step(states, x0, 100).numpy()
    out["cf/drift_t5"] = sde.drift(states, 5).numpy()
    out["cf/weights"] = sde.weights(tt).numpy()
    torch.manual_seed(5)
    ns = sde.noise_state(mu)
    torch.manual_seed(5)
    out["cf/noise_state_eps"] = torch.randn_like(mu).numpy()
    out["cf/noise_state"] = ns.numpy()

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, len(out), "arrays", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
