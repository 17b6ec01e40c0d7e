"""Pins oracle/sde_ref.py against golden vectors generated from the real reference
(utils/sde_utils.py, via tests/golden/make_golden_sde.py).  Bit-exact (same torch-CPU op order)."""
import os

import numpy as np

import pytest
import torch

from oracle import sde_ref

CFGS = {
    "cos100": dict(max_sigma=0.4, T=100, schedule="cosine", eps=0.01),
    "cos100_s50": dict(max_sigma=0.4, T=100, sample_T=50, schedule="cosine", eps=0.01),
    "cos1000": dict(max_sigma=0.4, T=1000, schedule="cosine", eps=0.01),
    "lin100": dict(max_sigma=0.4, T=100, schedule="linear", eps=0.01),
    "const100": dict(max_sigma=0.4, T=100, schedule="constant", eps=0.01),
    "cos100_ms50": dict(max_sigma=50, T=100, schedule="cosine", eps=0.01),
    "cos100_eps005": dict(max_sigma=0.25, T=100, schedule="cosine", eps=0.005),
}


def analytic_model(x, mu, t, **kw):
    return 0.3 * x - 0.2 * mu + (0.01 * float(t)) * (x * mu)


@pytest.mark.parametrize("name", list(CFGS))
def test_tables_bit_exact(golden_sde, name):
    tb = sde_ref.irsde_tables(**CFGS[name])
    for k in ["thetas", "sigmas", "thetas_cumsum", "sigma_bars"]:
        assert np.array_equal(tb[k].numpy(), golden_sde[f"{name}/{k}"]), k
    assert float(tb["dt"]) == float(golden_sde[f"{name}/dt"])
    assert tb["max_sigma"] == float(golden_sde[f"{name}/max_sigma"])
    assert tb["sample_scale"] == float(golden_sde[f"{name}/sample_scale"])


def test_survey_known_answers():
    # SURVEY.md §8c known answers captured from the reference
    tb = sde_ref.irsde_tables(0.4, T=100, schedule="cosine", eps=0.01)
    assert len(tb["thetas"]) == 101
    assert float(tb["dt"]) == pytest.approx(0.09047583490610123, abs=0)
    assert float(tb["thetas"][0]) == 6.142854690551758e-4
    assert float(tb["thetas"][-1]) == 0.9997665882110596
    assert float(tb["sigma_bars"][-1]) == 0.39997997879981995
    tb = sde_ref.irsde_tables(0.4, T=100, sample_T=50)
    assert len(tb["thetas"]) == 51 and tb["sample_scale"] == 2.0
    tb = sde_ref.irsde_tables(50, T=100)
    assert tb["max_sigma"] == pytest.approx(0.196078, abs=1e-6)


@pytest.mark.parametrize("tag,cfg,nsteps", [("t8", "cos100", 3), ("t64", "cos100_s50", 3), ("t8full", "cos100_s50", 50)])
def test_trajectories_bit_exact(golden_sde, tag, cfg, nsteps):
    g = golden_sde
    sde = sde_ref.IRSDERef(**CFGS[cfg])
    mu = torch.from_numpy(g[f"{tag}/mu"])
    xT = torch.from_numpy(g[f"{tag}/xT"])
    noises = torch.from_numpy(g[f"{tag}/noises"])
    sde.set_mu(mu)
    sde.set_model(analytic_model)
    x = sde.reverse_sde(xT, noises, T=nsteps)
    assert np.array_equal(x.numpy(), g[f"{tag}/x_sde"])
    x = sde.reverse_ode(xT, T=nsteps)
    assert np.array_equal(x.numpy(), g[f"{tag}/x_ode"])
    # per-step pieces at the top of the schedule
    x = xT.clone()
    for i, t in enumerate(g[f"{tag}/top_ts"].tolist()):
        score = sde.score_fn(x, t, sde.sample_scale)
        xm = sde.reverse_sde_step_mean(x, score, t)
        xo = sde.reverse_ode_step(x, score, t)
        x = sde.reverse_sde_step(x, score, t, noises[i])
        ref = g[f"{tag}/top_steps"][i]
        assert np.array_equal(score.numpy(), ref[0])
        assert np.array_equal(xm.numpy(), ref[1])
        assert np.array_equal(xo.numpy(), ref[2])
        assert np.array_equal(x.numpy(), ref[3])


def test_state_sampler_and_closed_forms(golden_sde):
    g = golden_sde
    sde = sde_ref.IRSDERef(**CFGS["cos100"])
    x0 = torch.from_numpy(g["grs/x0"])
    mu = torch.from_numpy(g["grs/mu"])
    t = torch.from_numpy(g["grs/t"])
    eps = torch.from_numpy(g["grs/eps"])
    assert t.shape == (4, 1, 1, 1) and t.dtype == torch.int64
    _, states = sde.generate_random_states(x0, mu, t, eps)
    assert states.dtype == torch.float32
    assert np.array_equal(states.numpy(), g["grs/states"])
    tt = torch.from_numpy(g["cf/t"])
    assert np.array_equal(sde.mu_bar(x0, tt).numpy(), g["cf/mu_bar"])
    assert np.array_equal(sde.get_real_noise(states, x0, tt).numpy(), g["cf/real_noise"])
    assert np.array_equal(sde.get_real_score(states, x0, tt).numpy(), g["cf/real_score"])
    assert np.array_equal(sde.get_init_state_from_noise(states, eps, tt).numpy(), g["cf/init_from_noise"])
    assert np.array_equal(sde.reverse_optimum_step(states, x0, 37).numpy(), g["cf/optimum_t37"])
    assert np.array_equal(sde.reverse_optimum_step(states, x0, 100).numpy(), g["cf/optimum_t100"])
    assert np.array_equal(sde.drift(states, 5).numpy(), g["cf/drift_t5"])
    assert np.array_equal(sde.weights(tt).numpy(), g["cf/weights"])
    ns = sde.noise_state(mu, torch.from_numpy(g["cf/noise_state_eps"]))
    assert np.array_equal(ns.numpy(), g["cf/noise_state"])


def test_sample_T_lt_T_indexerror_quirk():
    # tables have sample_T+1 rows, so t in (sample_T, T] is out of range (reference
    # raises IndexError in generate_random_states when sample_T < T).
    sde = sde_ref.IRSDERef(0.4, T=100, sample_T=50)
    with pytest.raises(IndexError):
        sde.sigma_bar(torch.tensor([[[[88]]]]))


def test_drift_sde_spec_identities():
    T = 100
    sde = sde_ref.DriftSDERef(T, None, None, max_sigma=0.4)
    assert float(sde.drift_schedule[0]) == 0.0 and float(sde.drift_schedule[T]) == 1.0
    assert torch.all(sde.drift_schedule[1:] >= sde.drift_schedule[:-1])
    g = torch.Generator().manual_seed(0)
    x0 = torch.rand(3, 1, 8, 8, generator=g) * 2 - 1
    cond = torch.rand(3, 1, 8, 8, generator=g) * 2 - 1
    eps = torch.randn(3, 1, 8, 8, generator=g)
    t = torch.tensor([1, 50, 100]).reshape(3, 1, 1, 1)
    tt, x_t, drift, std_noise, noise = sde.forward_diffusion(x0, cond, t, eps)
    # drift_noise_model.py:492: x0 + drift_schedule[t]*(cond-x0); :585: max_sigma*sqrt(noise_schedule[t])*eps == noise
    assert torch.allclose(x_t - noise, x0 + sde.drift_schedule[t] * (cond - x0), atol=1e-6)
    assert torch.allclose(noise, 0.4 * torch.sqrt(sde.noise_schedule[t]) * std_noise)
    # at t=T the state is cond + max_sigma*eps
    assert torch.allclose(x_t[2], cond[2] + 0.4 * eps[2], atol=1e-6)
    # with oracle (perfect) networks and eta=0 the reverse chain recovers x0 exactly-ish
    sde0 = sde_ref.DriftSDERef(T, lambda a, b, t, n, te, image_context=None: cond - x0,
                               lambda a, b, t, n, te, image_context=None: eps, max_sigma=0.4, eta=0.0)
    xT = cond + 0.4 * eps
    out = sde0.reverse_ddpm(cond, None, None, xT, [torch.zeros_like(x0)] * T)
    assert torch.allclose(out, x0, atol=2e-5)


# ---- second fixture (tests/golden/make_golden_sde2.py): the full method surface, scalar and per-sample t, mu tensor and mu = 0. ----
SURFACE = ["mu_bar", "drift", "sde_reverse_drift", "ode_reverse_drift", "dispersion", "score_from_noise", "forward_step",
           "reverse_sde_step_mean", "reverse_sde_step", "reverse_ode_step", "real_noise", "real_score", "init_from_noise",
           "reverse_optimum_step", "weights"]


def surface_call(sde, name, i, t, inject):
    """One method of the IRSDE surface on the golden2 inputs `i`; inject(z) -> the kwargs that carry the draw."""
    x, x0, score, noise, z = i["x"], i["x0"], i["score"], i["noise"], i["z"]
    return {
        "mu_bar": lambda: sde.mu_bar(x0, t),
        "drift": lambda: sde.drift(x, t),
        "sde_reverse_drift": lambda: sde.sde_reverse_drift(x, score, t),
        "ode_reverse_drift": lambda: sde.ode_reverse_drift(x, score, t),
        "dispersion": lambda: sde.dispersion(x, t, *inject(z)[0], **inject(z)[1]),
        "score_from_noise": lambda: sde.get_score_from_noise(noise, t),
        "forward_step": lambda: sde.forward_step(x, t, *inject(z)[0], **inject(z)[1]),
        "reverse_sde_step_mean": lambda: sde.reverse_sde_step_mean(x, score, t),
        "reverse_sde_step": lambda: sde.reverse_sde_step(x, score, t, *inject(z)[0], **inject(z)[1]),
        "reverse_ode_step": lambda: sde.reverse_ode_step(x, score, t),
        "real_noise": lambda: sde.get_real_noise(x, x0, t),
        "real_score": lambda: sde.get_real_score(x, x0, t),
        "init_from_noise": lambda: sde.get_init_state_from_noise(x, noise, t),
        "reverse_optimum_step": lambda: sde.reverse_optimum_step(x, x0, t),
        "weights": lambda: sde.weights(t),
    }[name]()


@pytest.mark.parametrize("mu_tag", ["mu", "mu0"])
@pytest.mark.parametrize("tname", ["t1", "t42", "t100", "tt"])
def test_method_surface_bit_exact(golden_sde2, mu_tag, tname):
    g = golden_sde2
    i = {k: torch.from_numpy(g[f"in/{k}"]) for k in ["x", "x0", "mu", "score", "noise", "z"]}
    sde = sde_ref.IRSDERef(**CFGS["cos100"])
    sde.set_mu(i["mu"] if mu_tag == "mu" else 0.)
    t = torch.from_numpy(g["in/tt"]) if tname == "tt" else int(tname[1:])
    for name in SURFACE:
        key = f"{mu_tag}/{tname}/{name}"
        if key not in g.files:
            assert name == "reverse_optimum_step" and tname == "tt"
            continue
        out = surface_call(sde, name, i, t, lambda z: ((z,), {}))
        assert np.array_equal(out.numpy(), g[key]), key


def test_optimal_reverse_and_x0_score(golden_sde2):
    g = golden_sde2
    i = {k: torch.from_numpy(g[f"in/{k}"]) for k in ["x", "x0", "mu"]}
    sde = sde_ref.IRSDERef(**CFGS["cos100"])
    sde.set_mu(i["mu"])
    assert np.array_equal(sde.optimal_reverse(i["x"], i["x0"], T=7).numpy(), g["opt/optimal_reverse_T7"])
    sde.set_model(lambda xx, m, t, **kw: 0.8 * xx + 0.1 * m)
    assert np.array_equal(sde.score_fn_(i["x"], 9, 1.0).numpy(), g["opt/score_fn_x0pred_t9"])


@pytest.mark.parametrize("T", [50, 100, 1000])
def test_cosine_drift_level_increments_match_the_reference_function(T):
    """`get_drift_deferential_cosine(t, T)` (models/drift_noise_model.py:10-16) -- the reference-held piece of the drift schedule:
    level(t+1) - level(t) of the half-cosine level.  Golden increments come from the reference's own function definition
    (tests/golden/make_golden_drift_cosine.py); the oracle's and the product's `cosine` level tables must reproduce them."""
    import numpy as np
    from instancediff_amd.models.SDEs.driftSDE import _level_table
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "drift_cosine_golden.npz"))
    inc = gold[f"T{T}/increment"]
    assert np.array_equal(inc, gold[f"T{T}/increment_int_t"]) and inc.shape == (T,)
    assert abs(inc.sum() - 1.0) < 1e-12   # the increments telescope to level(T) - level(0) = 1
    for name, table in (("oracle", sde_ref.drift_level_table(T, "cosine")), ("product", _level_table(T, "cosine"))):
        lv = table.double().numpy()
        assert lv.shape == (T + 1,) and lv[0] == 0.0 and lv[-1] == 1.0
        err = np.abs((lv[1:] - lv[:-1]) - inc).max()
        assert err < 1.5e-7, (name, err)  # the tables are stored in fp32: two roundings of values <= 1
    # and in fp64 the closed form itself is the reference's, to the last bits
    t = np.arange(T + 1, dtype=np.float64)
    lv64 = (1 - np.cos(t * np.pi / T)) / 2
    assert np.abs((lv64[1:] - lv64[:-1]) - inc).max() < 1e-15
