"""ORACLE (test infrastructure, not product code).

CPU restatement of the reference's mean-reverting SDE (`IRSDE`) and of the reconstructed
instance-wise drift SDE (`driftSDE`), in plain single-threaded torch-CPU fp32 tensor arithmetic with
the SAME operation order as the reference so results are bit-identical to it.

Pinned: every function here is checked against golden vectors produced by the real reference
(`tests/golden/make_golden_sde.py` imports /root/reference/utils/sde_utils.py) in
`tests/test_oracle_sde.py`.  driftSDE has no reference source -> "parity unpinned"
for that class; it restates the build's own frozen spec.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import torch


# ----------------------------------------------------------------------------------------------
# IRSDE schedule tables  (reference: utils/sde_utils.py:92-155)
# ----------------------------------------------------------------------------------------------
def irsde_tables(max_sigma, T=100, sample_T=-1, schedule="cosine", eps=0.01):
    """Returns dict(thetas, sigmas, thetas_cumsum, sigma_bars [sample_T+1 fp32], dt (0-dim fp32),
    max_sigma, sample_T, sample_scale).  Follows sde_utils.py:85-152 line by line."""
    max_sigma = max_sigma / 255 if max_sigma >= 1 else max_sigma  # :87
    sample_T = T if sample_T < 0 else sample_T  # :88
    sample_scale = T / sample_T  # :89
    n = sample_T
    if schedule == "cosine":  # :113-124
        timesteps = n + 2
        steps = timesteps + 1
        x = torch.linspace(0, timesteps, steps, dtype=torch.float32)
        ac = torch.cos(((x / timesteps) + 0.008) / (1 + 0.008) * math.pi * 0.5) ** 2
        ac = ac / ac[0]
        thetas = 1 - ac[1:-1]
    elif schedule == "linear":  # :102-111
        timesteps = n + 1
        scale = 1000 / timesteps
        thetas = torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float32)
    elif schedule == "constant":  # :94-100
        thetas = torch.ones(n + 1, dtype=torch.float32)
    else:
        raise ValueError(f"unknown schedule {schedule}")
    sigmas = torch.sqrt(max_sigma ** 2 * 2 * thetas)  # :130
    thetas_cumsum = torch.cumsum(thetas, dim=0) - thetas[0]  # :145
    dt = -1 / thetas_cumsum[-1] * math.log(eps)  # :146  (0-dim fp32 tensor)
    sigma_bars = torch.sqrt(max_sigma ** 2 * (1 - torch.exp(-2 * thetas_cumsum * dt)))  # :133
    return dict(thetas=thetas, sigmas=sigmas, thetas_cumsum=thetas_cumsum, sigma_bars=sigma_bars,
                dt=dt, max_sigma=max_sigma, sample_T=sample_T, sample_scale=sample_scale, T=T)


class IRSDERef:
    """Restatement of reference `IRSDE` (sde_utils.py:81-343) minus image dumping / scipy sampler."""

    def __init__(self, max_sigma, T=100, sample_T=-1, schedule="cosine", eps=0.01):
        tb = irsde_tables(max_sigma, T, sample_T, schedule, eps)
        self.__dict__.update(tb)
        self.mu = 0.0
        self.model = None

    def set_mu(self, mu):
        self.mu = mu

    def set_model(self, model):
        self.model = model

    # :169-173
    def mu_bar(self, x0, t):
        return self.mu + (x0 - self.mu) * torch.exp(-self.thetas_cumsum[t] * self.dt)

    def sigma_bar(self, t):
        return self.sigma_bars[t]

    # :175-188
    def drift(self, x, t):
        return self.thetas[t] * (self.mu - x) * self.dt

    def sde_reverse_drift(self, x, score, t):
        return (self.thetas[t] * (self.mu - x) - self.sigmas[t] ** 2 * score) * self.dt

    def ode_reverse_drift(self, x, score, t):
        return (self.thetas[t] * (self.mu - x) - 0.5 * self.sigmas[t] ** 2 * score) * self.dt

    def dispersion(self, x, t, noise):
        # reference draws torch.randn_like(x); the oracle takes the draw as an argument (:185)
        return self.sigmas[t] * (noise * math.sqrt(self.dt))

    def get_score_from_noise(self, noise, t):
        return -noise / self.sigma_bar(t)

    # :41-49
    def reverse_sde_step_mean(self, x, score, t):
        return x - self.sde_reverse_drift(x, score, t)

    def reverse_sde_step(self, x, score, t, noise):
        return x - self.sde_reverse_drift(x, score, t) - self.dispersion(x, t, noise)

    def reverse_ode_step(self, x, score, t):
        return x - self.ode_reverse_drift(x, score, t)

    def forward_step(self, x, t, noise):  # :38-39
        return x + self.drift(x, t) + self.dispersion(x, t, noise)

    def score_fn_(self, x, t, scale=1.0):  # :190-194 (x0-predicting model)
        x0 = self.model(x, self.mu, t * scale)
        return -(x - self.mu_bar(x0, t)) / self.sigma_bar(t) ** 2

    def optimal_reverse(self, xt, x0, T=-1):  # :308-314
        T = self.T if T < 0 else T
        x = xt.clone()
        for t in reversed(range(1, T + 1)):
            x = self.reverse_optimum_step(x, x0, t)
        return x

    def score_fn(self, x, t, scale=1.0, **kw):  # :196-199
        noise = self.model(x, self.mu, t * scale, **kw)
        return self.get_score_from_noise(noise, t)

    # :244-279 (loops) -- noises[i] is the draw used at loop iteration i (t = T, T-1, ...)
    def reverse_sde(self, xt, noises, T=-1, **kw):
        T = self.sample_T if T < 0 else T
        x = xt.clone()
        for i, t in enumerate(reversed(range(1, T + 1))):
            score = self.score_fn(x, t, self.sample_scale, **kw)
            x = self.reverse_sde_step(x, score, t, noises[i])
        return x

    def reverse_ode(self, xt, T=-1, **kw):
        T = self.sample_T if T < 0 else T
        x = xt.clone()
        for t in reversed(range(1, T + 1)):
            score = self.score_fn(x, t, self.sample_scale, **kw)
            x = self.reverse_ode_step(x, score, t)
        return x

    def reverse_mean(self, xt, T=-1, **kw):
        T = self.sample_T if T < 0 else T
        x = xt.clone()
        for t in reversed(range(1, T + 1)):
            score = self.score_fn(x, t, self.sample_scale, **kw)
            x = self.reverse_sde_step_mean(x, score, t)
        return x

    # :206-230
    def reverse_optimum_step(self, xt, x0, t):
        A = torch.exp(-self.thetas[t] * self.dt)
        B = torch.exp(-self.thetas_cumsum[t] * self.dt)
        C = torch.exp(-self.thetas_cumsum[t - 1] * self.dt)
        term1 = A * (1 - C ** 2) / (1 - B ** 2)
        term2 = C * (1 - A ** 2) / (1 - B ** 2)
        return term1 * (xt - self.mu) + term2 * (x0 - self.mu) + self.mu

    def get_real_noise(self, xt, x0, t):
        return (xt - self.mu_bar(x0, t)) / self.sigma_bar(t)

    def get_real_score(self, xt, x0, t):
        return -(xt - self.mu_bar(x0, t)) / self.sigma_bar(t) ** 2

    def get_init_state_from_noise(self, xt, noise, t):
        A = torch.exp(self.thetas_cumsum[t] * self.dt)
        return (xt - self.mu - self.sigma_bar(t) * noise) * A + self.mu

    def weights(self, t):
        return torch.exp(-self.thetas_cumsum[t] * self.dt)

    # :322-341 -- timesteps / eps are passed in (the reference draws them with torch.randint/randn_like)
    def generate_random_states(self, x0, mu, timesteps, eps):
        self.set_mu(mu)
        state_mean = self.mu_bar(x0, timesteps)
        noise_level = self.sigma_bar(timesteps)
        return timesteps, (eps * noise_level + state_mean).to(torch.float32)

    def noise_state(self, tensor, eps):
        return tensor + eps * self.max_sigma


# ----------------------------------------------------------------------------------------------
# driftSDE -- frozen build spec (no reference source; contract from drift_noise_model.py:190,357,
# 492,585,650 and config.yml:169-175).  "parity unpinned".
# ----------------------------------------------------------------------------------------------
def drift_level_table(T, kind):
    """level[t], t=0..T, level[0]=0, level[T]=1, fp32 (computed in fp64 then rounded)."""
    t = torch.arange(T + 1, dtype=torch.float64)
    if kind == "cosine":  # drift_noise_model.py:10-16: (1 - cos(pi t / T)) / 2
        lv = (1 - torch.cos(t * math.pi / T)) / 2
    elif kind == "sigmoid":
        k = 6.0
        s = torch.sigmoid(k * (2 * t / T - 1))
        s0, s1 = torch.sigmoid(torch.tensor(-k, dtype=torch.float64)), torch.sigmoid(torch.tensor(k, dtype=torch.float64))
        lv = (s - s0) / (s1 - s0)
    elif kind == "linear":
        lv = t / T
    else:
        raise ValueError(kind)
    lv[0] = 0.0
    lv[-1] = 1.0
    return lv.to(torch.float32)


def drift_step_coeffs(drift_schedule, noise_schedule, max_sigma, T, eta=1.0):
    """Per-step scalars (a_t, b_t, c_t), t=0..T (row 0 unused), fp32 from fp64 arithmetic:
       x_{t-1} = x_t - a_t * R_hat - b_t * eps_hat + c_t * z
       a_t = d_t - d_{t-1};  s_t = max_sigma*sqrt(n_t);  eta_t = eta * s_{t-1} * sqrt(1 - s_{t-1}^2/s_t^2)
       b_t = s_t - sqrt(s_{t-1}^2 - eta_t^2);  c_t = eta_t."""
    d = drift_schedule.to(torch.float64)
    s = max_sigma * torch.sqrt(noise_schedule.to(torch.float64))
    a = torch.zeros(T + 1, dtype=torch.float64)
    b = torch.zeros(T + 1, dtype=torch.float64)
    c = torch.zeros(T + 1, dtype=torch.float64)
    for t in range(1, T + 1):
        a[t] = d[t] - d[t - 1]
        ratio = (s[t - 1] / s[t]) ** 2 if s[t] > 0 else 0.0
        et = eta * s[t - 1] * math.sqrt(max(1.0 - float(ratio), 0.0))
        keep = math.sqrt(max(float(s[t - 1]) ** 2 - et ** 2, 0.0))
        b[t] = s[t] - keep
        c[t] = et
    return a.to(torch.float32), b.to(torch.float32), c.to(torch.float32)


class DriftSDERef:
    def __init__(self, T, drift_net, noise_net, max_sigma=0.4, drift_schedule="sigmoid", noise_schedule="sigmoid",
                 eta=1.0):
        self.T = T
        self.max_sigma = max_sigma
        self.drift_net = drift_net
        self.noise_net = noise_net
        self.drift_schedule = drift_level_table(T, drift_schedule)
        self.noise_schedule = drift_level_table(T, noise_schedule)
        self.a, self.b, self.c = drift_step_coeffs(self.drift_schedule, self.noise_schedule, max_sigma, T, eta)

    def forward_diffusion(self, x0, cond, t, eps):
        """(t, x_t, drift, std_noise, noise); t [B,1,1,1] long in [1,T], eps ~ N(0,1) passed in."""
        d = self.drift_schedule[t]
        n = self.noise_schedule[t]
        drift = d * (cond - x0)
        noise = (self.max_sigma * torch.sqrt(n)) * eps
        x_t = x0 + drift + noise
        return t, x_t, drift, eps, noise

    def reverse_ddpm(self, cond, names, text_encoder, x_T, noises, reverse_type="std", optimize_type="inputRes",
                     image_context=None):
        """x_T: initial state (cond + max_sigma*eps); noises[i]: draw used at loop iteration i (t=T..1)."""
        x = x_T.clone()
        B = cond.shape[0]
        for i, t in enumerate(reversed(range(1, self.T + 1))):
            tt = torch.full((B,), t, dtype=torch.long)
            rd = self.drift_net(x - cond, cond, tt, names, text_encoder, image_context=image_context)
            rn = self.noise_net(x - cond, x, tt, names, text_encoder, image_context=image_context)
            rd = rd[0] if isinstance(rd, tuple) else rd
            rn = rn[0] if isinstance(rn, tuple) else rn
            x = drift_reverse_update(x, rd, rn, noises[i], self.a[t], self.b[t], self.c[t])
        return x


def drift_reverse_update(x, r_hat, e_hat, z, a, b, c):
    """x - a*r_hat - b*e_hat + c*z with one rounding per op, left to right (fp32)."""
    return ((x - a * r_hat) - b * e_hat) + c * z


def psnr(pred, target):
    """PSNR on x/2+0.5, data_range 1 (trainUM.py:319-323 semantics), float64."""
    p = pred.to(torch.float64) / 2 + 0.5
    g = target.to(torch.float64) / 2 + 0.5
    mse = torch.mean((p - g) ** 2)
    return float(10.0 * torch.log10(1.0 / mse))
