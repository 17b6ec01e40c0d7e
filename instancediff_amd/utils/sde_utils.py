"""`SDE` / `IRSDE` -- the mean-reverting SDE with the reference's full method surface (utils/sde_utils.py:10-343),
every tensor operation running in gfx950 kernels behind the C ABI (idiff_irsde_map, idiff_irsde_reverse_step).

Contract kept from the reference, method by method (same names, argument order and meaning):
  * the step functions take the SCORE (`reverse_sde_step(x, score, t)`, sde_utils.py:41-49) -- the reference's literal
    loop body `score = sde.score_fn(x, t, scale); x = sde.reverse_sde_step(x, score, t)` (:249-250) runs unchanged;
  * `t` may be a python int (the loops) or a long tensor [B,1,1,1] (training-state sampler, closed forms), as the
    reference's table indexing `self.thetas[t]` allows.
Each method is ONE kernel launch that reproduces the reference's fp32 rounding sequence (no FMA contraction, IEEE
quotients), so given the same draws the results are bit-identical to the CPU reference (tests/golden/).

Schedule tables are built exactly as the reference builds them -- on the host with fp32 torch ops, then moved to the
device (sde_utils.py:92-152 does `.to(self.device)` on CPU-built tables).  Per-step scalars are read from the host
copies, so no method synchronises with the device (a `t` tensor that lives on the GPU is the one exception).

Extra to the reference surface: `reverse_sde_step_from_noise` (and `_mean`/`_ode` forms) -- score_fn's -noise/sigma_bar
folded into the step kernel; the `reverse_sde` / `reverse_ode` loops use it (1 launch per step instead of 2, same bits);
optional `z=` / `eps=` / `noises=` arguments inject the Gaussian draws the reference takes from torch.randn_like
(device-dependent stream): parity runs inject, throughput runs use on-device Philox4x32-10 keyed by (seed, call count).

Out of scope: `ode_sampler` (scipy RK45, "not used", :282-306) and the image dumping of
`forward` / `reverse_sde(save_states=True)` (:239-241,253-259).
"""
import abc
import math

import torch

from .. import ops


class SDE(abc.ABC):
    """Abstract interface of sde_utils.py:10-75 (the composition rules of the step functions live here)."""

    def __init__(self, T, device=None):
        self.T = T
        self.dt = 1 / T
        self.device = device

    @abc.abstractmethod
    def drift(self, x, t):
        pass

    @abc.abstractmethod
    def dispersion(self, x, t):
        pass

    @abc.abstractmethod
    def sde_reverse_drift(self, x, score, t):
        pass

    @abc.abstractmethod
    def ode_reverse_drift(self, x, score, t):
        pass

    @abc.abstractmethod
    def score_fn(self, x, t):
        pass


class IRSDE(SDE):
    """Let timestep t run from 1 to T; state t=0 is never used (sde_utils.py:82-84)."""

    def __init__(self, max_sigma, T=100, sample_T=-1, schedule='cosine', eps=0.01, device=None):
        super().__init__(T, device)
        self.max_sigma = max_sigma / 255 if max_sigma >= 1 else max_sigma  # :87
        self.sample_T = self.T if sample_T < 0 else sample_T  # :88
        self.sample_scale = self.T / self.sample_T  # :89
        self._initialize(self.max_sigma, self.sample_T, schedule, eps)
        self.seed = 0
        self._noise_off = 0

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
        # host copies drive the kernels' scalar arguments; device copies serve the indexable tables
        self._h = dict(thetas=thetas, sigmas=sigmas, thetas_cumsum=thetas_cumsum, sigma_bars=sigma_bars)
        self._dt = float(self.dt)
        self._sqrt_dt = math.sqrt(self.dt)  # :185 math.sqrt(self.dt); the tensor op then rounds it to fp32
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

    #####################################
    def set_mu(self, mu):  # :160
        self.mu = mu

    def set_model(self, model):  # :164
        self.model = model

    def set_seed(self, seed):
        """key of the on-device Philox stream that stands in for torch.randn_like"""
        self.seed = int(seed)
        self._noise_off = 0

    def _next_offset(self, numel):
        """Philox counter offset of the next draw of `numel` normals (4 per counter).  Cumulative: successive draws use disjoint
        counter ranges whatever their sizes (a per-call index times the CURRENT size re-read counters an earlier, larger draw had
        consumed); same-shape sequences are reproducible as before."""
        off = self._noise_off
        self._noise_off += (numel + 3) // 4
        return off

    # ---- coefficient plumbing ------------------------------------------------------------------------
    def _rows(self, t):
        """host LongTensor of table rows for `t` (python int -> None: scalar coefficients travel by value)"""
        if torch.is_tensor(t):
            if t.numel() == 1 and t.dim() == 0:
                return None
            return t.detach().to('cpu').reshape(-1).long()
        return None

    def _coef(self, like, t, fn):
        """fn(row index: int or LongTensor) -> list of up to 6 fp32 host values/tensors.  Returns the kwargs of ops.irsde_map:
        k=[...] for a scalar t, coef_dev=[B,6] (one row per sample) for a [B,...] tensor t."""
        rows = self._rows(t)
        if rows is None:
            return dict(k=[float(v) for v in fn(int(t))])
        B = like.shape[0]
        if rows.numel() != B:
            raise ValueError(f"t has {rows.numel()} entries for a batch of {B}")
        if int(rows.max()) >= self._h["thetas"].numel() or int(rows.min()) < -self._h["thetas"].numel():
            raise IndexError("timestep outside the schedule tables (sample_T < T: reference quirk, sde_utils.py:330-335)")
        vals = fn(rows)
        tab = torch.zeros((B, 6), dtype=torch.float32)
        for i, v in enumerate(vals):
            tab[:, i] = v if torch.is_tensor(v) else float(v)
        return dict(coef_dev=tab.to(like.device, non_blocking=True))

    def _mu_for(self, like):
        return self.mu if torch.is_tensor(self.mu) else float(self.mu)

    def _draw_args(self, like, z):
        """(z, seed, offset): injected draws, or the next slice of the Philox stream"""
        if z is not None:
            return dict(z=z.contiguous())
        return dict(z=None, seed=self.seed, offset=self._next_offset(like.numel()))

    def _w(self, r):  # exp(-thetas_cumsum[t] * dt), fp32 on the host exactly as :170
        return torch.exp(-self._h["thetas_cumsum"][r] * self.dt)

    # ---- table lookups (:169-173, 216-220, 316-317) --------------------------------------------------
    def sigma_bar(self, t):
        return self.sigma_bars[t]

    def sigma(self, t):
        return self.sigmas[t]

    def theta(self, t):
        return self.thetas[t]

    def weights(self, t):
        """exp(-thetas_cumsum[t] * dt) (:316-317): a table lookup, formed on the host tables and moved to the device"""
        rows = self._rows(t)
        w = self._w(int(t) if rows is None else rows.reshape(t.shape))
        return w.to(self.device)

    # ---- closed forms --------------------------------------------------------------------------------
    def mu_bar(self, x0, t):  # :169-170
        x0 = x0.contiguous()
        return ops.irsde_map(ops.IRSDE_MU_BAR, x0, a=x0, mu=self._mu_for(x0), **self._coef(x0, t, lambda r: [self._w(r)]))

    def drift(self, x, t):  # :175-176
        x = x.contiguous()
        return ops.irsde_map(ops.IRSDE_DRIFT, x, a=x, mu=self._mu_for(x), **self._coef(x, t, lambda r: [self._h["thetas"][r], self.dt]))

    def _rev_coef(self, r, half):
        h = self._h
        s2 = h["sigmas"][r] ** 2
        return [h["thetas"][r], 0.5 * s2 if half else s2, self.dt, h["sigmas"][r], self._sqrt_dt]

    def sde_reverse_drift(self, x, score, t):  # :178-179
        x = x.contiguous()
        return ops.irsde_map(ops.IRSDE_REV_DRIFT, x, a=x, b=score.contiguous(), mu=self._mu_for(x), **self._coef(x, t, lambda r: self._rev_coef(r, False)))

    def ode_reverse_drift(self, x, score, t):  # :181-182
        x = x.contiguous()
        return ops.irsde_map(ops.IRSDE_REV_DRIFT, x, a=x, b=score.contiguous(), mu=self._mu_for(x), **self._coef(x, t, lambda r: self._rev_coef(r, True)))

    def dispersion(self, x, t, z=None):  # :184-185  sigma_t * (randn_like(x) * sqrt(dt))
        x = x.contiguous()
        return ops.irsde_map(ops.IRSDE_DISPERSION, x, **self._draw_args(x, z), **self._coef(x, t, lambda r: [self._h["sigmas"][r], self._sqrt_dt]))

    def get_score_from_noise(self, noise, t):  # :187-188
        noise = noise.contiguous()
        return ops.irsde_map(ops.IRSDE_SCORE_FROM_NOISE, noise, a=noise, **self._coef(noise, t, lambda r: [self._h["sigma_bars"][r]]))

    def get_real_noise(self, xt, x0, t):  # :222-223
        xt = xt.contiguous()
        return ops.irsde_map(ops.IRSDE_REAL_NOISE, xt, a=xt, b=x0.contiguous(), mu=self._mu_for(xt),
                             **self._coef(xt, t, lambda r: [self._w(r), self._h["sigma_bars"][r]]))

    def get_real_score(self, xt, x0, t):  # :225-226
        xt = xt.contiguous()
        return ops.irsde_map(ops.IRSDE_REAL_SCORE, xt, a=xt, b=x0.contiguous(), mu=self._mu_for(xt),
                             **self._coef(xt, t, lambda r: [self._w(r), self._h["sigma_bars"][r] ** 2]))

    def get_init_state_from_noise(self, xt, noise, t):  # :228-230
        xt = xt.contiguous()
        return ops.irsde_map(ops.IRSDE_INIT_FROM_NOISE, xt, a=xt, b=noise.contiguous(), mu=self._mu_for(xt),
                             **self._coef(xt, t, lambda r: [self._h["sigma_bars"][r], torch.exp(self._h["thetas_cumsum"][r] * self.dt)]))

    def reverse_optimum_step(self, xt, x0, t):  # :206-214
        def terms(r):
            h = self._h
            A = torch.exp(-h["thetas"][r] * self.dt)
            B = torch.exp(-h["thetas_cumsum"][r] * self.dt)
            C = torch.exp(-h["thetas_cumsum"][r - 1] * self.dt)
            return [A * (1 - C ** 2) / (1 - B ** 2), C * (1 - A ** 2) / (1 - B ** 2)]
        xt = xt.contiguous()
        return ops.irsde_map(ops.IRSDE_OPT_STEP, xt, a=xt, b=x0.contiguous(), mu=self._mu_for(xt), **self._coef(xt, t, terms))

    # ---- step functions: the composition rules of the SDE base class (:38-49), one launch each ------------
    def forward_step(self, x, t, z=None):  # :38-39  x + drift + dispersion
        x = x.contiguous()
        return ops.irsde_map(ops.IRSDE_FORWARD_STEP, x, a=x, mu=self._mu_for(x), **self._draw_args(x, z),
                             **self._coef(x, t, lambda r: [self._h["thetas"][r], self.dt, 0.0, self._h["sigmas"][r], self._sqrt_dt]))

    def reverse_sde_step_mean(self, x, score, t):  # :41-42
        x = x.contiguous()
        return ops.irsde_map(ops.IRSDE_STEP_MEAN, x, a=x, b=score.contiguous(), mu=self._mu_for(x), **self._coef(x, t, lambda r: self._rev_coef(r, False)))

    def reverse_sde_step(self, x, score, t, z=None):  # :45-46
        x = x.contiguous()
        return ops.irsde_map(ops.IRSDE_STEP_SDE, x, a=x, b=score.contiguous(), mu=self._mu_for(x), **self._draw_args(x, z),
                             **self._coef(x, t, lambda r: self._rev_coef(r, False)))

    def reverse_ode_step(self, x, score, t):  # :48-49
        x = x.contiguous()
        return ops.irsde_map(ops.IRSDE_STEP_MEAN, x, a=x, b=score.contiguous(), mu=self._mu_for(x), **self._coef(x, t, lambda r: self._rev_coef(r, True)))

    # ---- model evaluation (:190-203) ---------------------------------------------------------------------
    def score_fn_(self, x, t, scale=1.0):  # :190-194, x0-predicting model
        x0 = self.model(x, self.mu, t * scale)
        return self.get_real_score(x, x0, t)

    def score_fn(self, x, t, scale=1.0, **kwargs):  # :196-199
        noise = self.model(x, self.mu, t * scale, **kwargs)
        if isinstance(noise, tuple):  # the UNet returns (pred, [score maps]) with text_module == 'scoremap'
            noise = noise[0]
        return self.get_score_from_noise(noise, t)

    def noise_fn(self, x, t, scale=1.0, **kwargs):  # :201-203
        return self.model(x, self.mu, t * scale, **kwargs)

    # ---- fused fast path: -noise/sigma_bar folded into the step (same bits as score_fn + step) ------------
    def _step_from_noise(self, x, noise_pred, t, mode, z=None, out=None):
        h = self._h
        mu = self.mu if torch.is_tensor(self.mu) else torch.full_like(x, float(self.mu))
        off = 0
        if mode == ops.SDE_STEP and z is None:
            off = self._next_offset(x.numel())
        return ops.irsde_reverse_step(x, mu.contiguous(), noise_pred.contiguous(), z, float(h["thetas"][t]), float(h["sigmas"][t]),
                                      float(h["sigma_bars"][t]), self._dt, self._sqrt_dt, mode=mode, seed=self.seed, offset=off, out=out)

    def reverse_sde_step_from_noise(self, x, noise_pred, t, z=None):
        """reverse_sde_step(x, get_score_from_noise(noise_pred, t), t) in one launch"""
        return self._step_from_noise(x.contiguous(), noise_pred, t, ops.SDE_STEP, z)

    def reverse_sde_step_mean_from_noise(self, x, noise_pred, t):
        return self._step_from_noise(x.contiguous(), noise_pred, t, ops.SDE_MEAN)

    def reverse_ode_step_from_noise(self, x, noise_pred, t):
        return self._step_from_noise(x.contiguous(), noise_pred, t, ops.SDE_ODE)

    # ---- loops (:232-279, 308-314) -----------------------------------------------------------------------
    def forward(self, x0, T=-1, save_dir='forward_state', noises=None):  # :232-242 (image dumping out of scope)
        T = self.T if T < 0 else T
        x = x0.contiguous().clone()
        for i, t in enumerate(range(1, T + 1)):
            x = self.forward_step(x, t, None if noises is None else noises[i])
        return x

    @torch.no_grad()
    def _loop(self, xt, T, mode, noises=None, **kwargs):
        T = self.sample_T if T < 0 else T
        x = xt.contiguous().clone()
        for i, t in enumerate(reversed(range(1, T + 1))):
            noise = self.noise_fn(x, t, self.sample_scale, **kwargs)
            if isinstance(noise, tuple):
                noise = noise[0]
            z = None if noises is None else noises[i].contiguous()
            x = self._step_from_noise(x, noise, t, mode, z)
        return x

    def reverse_sde(self, xt, T=-1, save_states=False, save_dir='sde_state', noises=None, **kwargs):
        """Reverse SDE Euler loop (:244-261).  `noises` (optional [steps,...]) injects the draws (parity runs);
        otherwise the kernel draws Philox normals.  Image dumping (save_states) is out of scope."""
        return self._loop(xt, T, ops.SDE_STEP, noises, **kwargs)

    def reverse_ode(self, xt, T=-1, save_states=False, save_dir='ode_state', **kwargs):  # :263-279
        return self._loop(xt, T, ops.SDE_ODE, None, **kwargs)

    def reverse_mean(self, xt, T=-1, **kwargs):
        """the loop with reverse_sde_step_mean (the reference keeps it as a commented alternative, :251)"""
        return self._loop(xt, T, ops.SDE_MEAN, None, **kwargs)

    def optimal_reverse(self, xt, x0, T=-1):  # :308-314
        T = self.T if T < 0 else T
        x = xt.contiguous().clone()
        for t in reversed(range(1, T + 1)):
            x = self.reverse_optimum_step(x, x0, t)
        return x

    def ode_sampler(self, *a, **k):
        raise NotImplementedError("IRSDE.ode_sampler (scipy RK45 black-box solver, 'not used', sde_utils.py:282-306) is out of scope")

    # ---- training-state sampler (:322-341) ---------------------------------------------------------------
    def generate_random_states(self, x0, mu, timesteps=None, T_start=1, T_end=-1, eps=None):
        """noises * sigma_bar[t] + (mu + (x0-mu)*exp(-thetas_cumsum[t]*dt)) in the reference's rounding order (:333-336).
        timesteps drawn on the host with torch.randint like the reference unless given; eps drawn on-device unless given."""
        x0 = x0.to(self.device).contiguous()
        mu = mu.to(self.device).contiguous()
        self.set_mu(mu)
        if timesteps is None:
            batch = x0.shape[0]
            T_end = self.T + 1 if T_end <= 1 else T_end + 1
            timesteps = torch.randint(T_start, T_end, (batch, 1, 1, 1)).long()
        coef = self._coef(x0, timesteps, lambda r: [self._w(r), self._h["sigma_bars"][r]])
        noisy_states = ops.irsde_map(ops.IRSDE_RANDOM_STATES, x0, a=x0, mu=mu, **self._draw_args(x0, eps), **coef)
        return timesteps.to(self.device), noisy_states

    def _randn_like(self, x):
        return ops.randn(x.shape, x.device, self.seed, self._next_offset(x.numel()))

    def noise_state(self, tensor, eps=None):
        """tensor + randn * max_sigma (:340-341)."""
        tensor = tensor.contiguous()
        if eps is None:
            eps = self._randn_like(tensor)
        return ops.axpby(tensor, eps.contiguous(), 1.0, float(self.max_sigma))
