"""`IRSDE` -- mean-reverting SDE with the reference's interface (utils/sde_utils.py:81-343), the
reverse loops running on the fused gfx950 step kernel (idiff_irsde_reverse_step).

Schedule tables are built exactly as the reference builds them -- on the host with fp32 torch ops, then
moved to the device (sde_utils.py:92-152 does `.to(self.device)` on CPU-built tables) -- so they are
bit-identical to the reference's.  Everything per step (score_fn's -noise/sigma_bar, the reverse drift,
the dispersion and the noise draw) is ONE kernel launch instead of the reference's 5-6 ATen launches +
randn_like; per-step scalars (theta_t, sigma_t, sigma_bar_t, dt) are read from host copies of the tables,
so the loop has no device->host synchronisation.
"""
import math

import torch

from .. import ops


class IRSDE:
    """Let timestep t run from 1 to T; state t=0 is never used (sde_utils.py:82-84)."""

    def __init__(self, max_sigma, T=100, sample_T=-1, schedule='cosine', eps=0.01, device=None):
        self.T = T
        self.device = device
        self.max_sigma = max_sigma / 255 if max_sigma >= 1 else max_sigma  # :87
        self.sample_T = self.T if sample_T < 0 else sample_T  # :88
        self.sample_scale = self.T / self.sample_T  # :89
        self._initialize(self.max_sigma, self.sample_T, schedule, eps)
        self.seed = 0
        self._noise_calls = 0

    # ---- host-side schedule construction, operation order of sde_utils.py:94-147 -------------------
    def _initialize(self, max_sigma, T, schedule, eps=0.01):
        if schedule == 'cosine':  # :113-124
            timesteps = T + 2
            steps = timesteps + 1
            x = torch.linspace(0, timesteps, steps, dtype=torch.float32)
            alphas_cumprod = torch.cos(((x / timesteps) + 0.008) / (1 + 0.008) * math.pi * 0.5) ** 2
            alphas_cumprod = alphas_cumprod / alphas_cumprod[0]
            thetas = 1 - alphas_cumprod[1:-1]
        elif schedule == 'linear':  # :102-111
            timesteps = T + 1
            scale = 1000 / timesteps
            thetas = torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float32)
        elif schedule == 'constant':  # :94-100
            thetas = torch.ones(T + 1, dtype=torch.float32)
        else:
            raise ValueError(f'Not implemented such schedule yet!!! ({schedule})')  # reference prints and crashes later (:142)
        sigmas = torch.sqrt(max_sigma ** 2 * 2 * thetas)
        thetas_cumsum = torch.cumsum(thetas, dim=0) - thetas[0]
        self.dt = -1 / thetas_cumsum[-1] * math.log(eps)  # 0-dim fp32 tensor, as in the reference (:146)
        sigma_bars = torch.sqrt(max_sigma ** 2 * (1 - torch.exp(-2 * thetas_cumsum * self.dt)))
        # host copies drive the kernel's scalar arguments; device copies serve indexable tables
        self._h = dict(thetas=thetas, sigmas=sigmas, thetas_cumsum=thetas_cumsum, sigma_bars=sigma_bars)
        self._dt = float(self.dt)
        self._sqrt_dt = math.sqrt(self._dt)  # reference: math.sqrt(self.dt) (:185), then cast to fp32 by the tensor op
        self.thetas = thetas.to(self.device)
        self.sigmas = sigmas.to(self.device)
        self.thetas_cumsum = thetas_cumsum.to(self.device)
        self.sigma_bars = sigma_bars.to(self.device)
        self.mu = 0.
        self.model = None

    def set_gpu(self, device):
        self.device = device
        for k in ("thetas", "sigmas", "thetas_cumsum", "sigma_bars"):
            setattr(self, k, self._h[k].to(device))

    def set_mu(self, mu):
        self.mu = mu

    def set_model(self, model):
        self.model = model

    def set_seed(self, seed):
        self.seed = int(seed)
        self._noise_calls = 0

    # ---- closed forms used by training-state sampling (:169-173, 322-341) --------------------------
    def sigma_bar(self, t):
        return self.sigma_bars[t]

    def sigma(self, t):
        return self.sigmas[t]

    def theta(self, t):
        return self.thetas[t]

    def _coef(self, t_host, fn):
        """per-sample fp32 coefficients computed on the host from the host tables (t_host: LongTensor on CPU)."""
        return fn(t_host.reshape(-1)).to(torch.float32)

    def generate_random_states(self, x0, mu, timesteps=None, T_start=1, T_end=-1, eps=None):
        """x_t = mu + (x0-mu)*exp(-thetas_cumsum[t]*dt) + sigma_bar[t]*eps  (:322-338).  timesteps drawn on the
        host with torch.randint like the reference unless given; eps drawn on-device (Philox) unless given."""
        x0 = x0.to(self.device).contiguous()
        mu = mu.to(self.device).contiguous()
        self.set_mu(mu)
        B = x0.shape[0]
        if timesteps is None:
            T_end = self.T + 1 if T_end <= 1 else T_end + 1
            timesteps = torch.randint(T_start, T_end, (B, 1, 1, 1)).long()
        th = timesteps.detach().cpu().reshape(-1)
        if int(th.max()) >= self._h["thetas"].numel():
            raise IndexError("timestep outside the schedule tables (sample_T < T: reference quirk, sde_utils.py:330-335)")
        w = torch.exp(-self._h["thetas_cumsum"][th] * self.dt)  # fp32, host
        sb = self._h["sigma_bars"][th]
        if eps is None:
            eps = self._randn_like(x0)
        c0 = w.to(self.device)
        c1 = (1 - w).to(self.device)
        c2 = sb.to(self.device)
        states = ops.mix3_per_sample(x0, mu, eps.contiguous(), c0.contiguous(), c1.contiguous(), c2.contiguous())
        return timesteps.to(self.device), states

    def _randn_like(self, x):
        off = self._noise_calls * ((x.numel() + 3) // 4)
        self._noise_calls += 1
        return ops.randn(x.shape, x.device, self.seed, off)

    def noise_state(self, tensor, eps=None):
        """tensor + randn * max_sigma (:340-341)."""
        tensor = tensor.contiguous()
        if eps is None:
            eps = self._randn_like(tensor)
        return ops.axpby(tensor, eps.contiguous(), 1.0, float(self.max_sigma))

    # ---- model evaluation (:196-203) --------------------------------------------------------------
    def noise_fn(self, x, t, scale=1.0, **kwargs):
        return self.model(x, self.mu, t * scale, **kwargs)

    # ---- fused reverse steps ----------------------------------------------------------------------
    def _step(self, x, noise_pred, t, mode, z=None, out=None):
        h = self._h
        mu = self.mu if torch.is_tensor(self.mu) else torch.full_like(x, float(self.mu))
        off = 0
        if mode == ops.SDE_STEP and z is None:
            off = self._noise_calls * ((x.numel() + 3) // 4)
            self._noise_calls += 1
        return ops.irsde_reverse_step(x, mu, noise_pred.contiguous(), z, float(h["thetas"][t]), float(h["sigmas"][t]),
                                      float(h["sigma_bars"][t]), self._dt, self._sqrt_dt, mode=mode, seed=self.seed, offset=off, out=out)

    def reverse_sde_step(self, x, noise_pred, t, z=None):
        """x - sde_reverse_drift(x, -noise_pred/sigma_bar_t, t) - dispersion(x, t)  (:45-46,178-188) given the
        network's noise prediction (the score is formed inside the kernel)."""
        return self._step(x, noise_pred, t, ops.SDE_STEP, z)

    def reverse_sde_step_mean(self, x, noise_pred, t):
        return self._step(x, noise_pred, t, ops.SDE_MEAN)

    def reverse_ode_step(self, x, noise_pred, t):
        return self._step(x, noise_pred, t, ops.SDE_ODE)

    @torch.no_grad()
    def _loop(self, xt, T, mode, noises=None, **kwargs):
        T = self.sample_T if T < 0 else T
        x = xt.contiguous().clone()
        for i, t in enumerate(reversed(range(1, T + 1))):
            noise = self.noise_fn(x, t, self.sample_scale, **kwargs)
            if isinstance(noise, tuple):
                noise = noise[0]
            z = None if noises is None else noises[i].contiguous()
            x = self._step(x, noise, t, mode, z)
        return x

    def reverse_sde(self, xt, T=-1, save_states=False, save_dir='sde_state', noises=None, **kwargs):
        """Reverse SDE Euler loop (:244-261).  `noises` (optional [steps,...]) injects the draws (parity runs);
        otherwise the kernel draws Philox normals.  Image dumping (save_states) is out of scope."""
        return self._loop(xt, T, ops.SDE_STEP, noises, **kwargs)

    def reverse_ode(self, xt, T=-1, save_states=False, save_dir='ode_state', **kwargs):
        return self._loop(xt, T, ops.SDE_ODE, None, **kwargs)

    def reverse_mean(self, xt, T=-1, **kwargs):
        return self._loop(xt, T, ops.SDE_MEAN, None, **kwargs)
