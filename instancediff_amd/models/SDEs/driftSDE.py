"""`driftSDE` -- the instance-wise-drift diffusion of InstanceDiff on fused gfx950 kernels.

models/SDEs/driftSDE.py is absent from the reference snapshot; this class
implements the contract recoverable from its call sites:
  * forward_diffusion(x0, cond) -> (t, x_t, drift, std_noise, noise)      models/drift_noise_model.py:190
      x_t = x0 + drift_schedule[t]*(cond - x0) + max_sigma*sqrt(noise_schedule[t])*eps   (:492, :585)
  * reverse_ddpm(cond, names, text_encoder, reverse_type=, optimize_type=, image_context=) -> x0_hat   (:650)
  * indexable drift_schedule / noise_schedule, attrs T, max_sigma, set_gpu(device)   (:357,:490,:543; testUM.py:96)
  * ctor from `create_sde(nets, opt['sdes'][name])` with T, max_sigma, drift_schedule, noise_schedule
    (Configurations/config.yml:169-175; schedules 'sigmoid' | 'cosine' (drift_noise_model.py:10-16) | 'linear').
The reverse update is the build's frozen spec ("parity unpinned"):
    x_{t-1} = x_t - a_t*R_hat - b_t*eps_hat + c_t*z,   R_hat = drift_net(x_t-cond, cond, t), eps_hat = noise_net(x_t-cond, x_t, t)
with (a_t, b_t, c_t) from oracle-identical fp64 host arithmetic; ONE kernel per step does the update, draws z
(Philox) and emits the next step's `x_t - cond` network input.
"""
import math
import os

import torch

from ... import ops


def _level_table(T, kind):
    t = torch.arange(T + 1, dtype=torch.float64)
    if kind == "cosine":
        lv = (1 - torch.cos(t * math.pi / T)) / 2
    elif kind == "sigmoid":
        k = 6.0
        s = torch.sigmoid(k * (2 * t / T - 1))
        s0, s1 = torch.sigmoid(torch.tensor(-k, dtype=torch.float64)), torch.sigmoid(torch.tensor(k, dtype=torch.float64))
        lv = (s - s0) / (s1 - s0)
    elif kind == "linear":
        lv = t / T
    else:
        raise ValueError(f"unknown schedule '{kind}'")
    lv[0] = 0.0
    lv[-1] = 1.0
    return lv.to(torch.float32)


def _step_coeffs(d, n, max_sigma, T, eta):
    d = d.to(torch.float64)
    s = max_sigma * torch.sqrt(n.to(torch.float64))
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


class driftSDE:
    def __init__(self, nets=None, T=100, max_sigma=0.4, drift_schedule="sigmoid", noise_schedule="sigmoid", eta=1.0, device=None,
                 **_ignored):
        self.T = int(T)
        self.max_sigma = float(max_sigma)
        self.eta = float(eta)
        nets = nets or {}
        self.drift_net = nets.get("drift_net")
        self.noise_net = nets.get("noise_net")
        self._h_drift = _level_table(self.T, drift_schedule)
        self._h_noise = _level_table(self.T, noise_schedule)
        self._a, self._b, self._c = _step_coeffs(self._h_drift, self._h_noise, self.max_sigma, self.T, self.eta)
        self.device = device
        self.drift_schedule = self._h_drift.to(device) if device is not None else self._h_drift
        self.noise_schedule = self._h_noise.to(device) if device is not None else self._h_noise
        self.seed = 0
        self._calls = 0
        self.two_streams = bool(int(os.environ.get("IDIFF_TWO_STREAMS", "1")))
        self._streams = None

    def set_gpu(self, device):
        self.device = device
        self.drift_schedule = self._h_drift.to(device)
        self.noise_schedule = self._h_noise.to(device)

    def set_seed(self, seed):
        self.seed = int(seed)
        self._calls = 0

    def _randn_like(self, x):
        off = self._calls * ((x.numel() + 3) // 4)
        self._calls += 1
        return ops.randn(x.shape, x.device, self.seed, off)

    # ---- training-state sampler -------------------------------------------------------------------
    def forward_diffusion(self, x0, cond, t=None, eps=None):
        """-> (t [B,1,1,1] long, x_t, drift, std_noise, noise); t drawn on the host like the reference's samplers
        (utils/sde_utils.py:330-331), eps on-device unless injected."""
        x0 = x0.contiguous()
        cond = cond.contiguous()
        dev = x0.device
        B = x0.shape[0]
        if t is None:
            t = torch.randint(1, self.T + 1, (B, 1, 1, 1)).long()
        th = t.detach().cpu().reshape(-1)
        d = self._h_drift[th]
        sn = (self.max_sigma * torch.sqrt(self._h_noise[th])).to(torch.float32)
        if eps is None:
            eps = self._randn_like(x0)
        eps = eps.contiguous()
        zero = torch.zeros(B, dtype=torch.float32)
        dd, nd, sd, zd, od = d.to(dev), (-d).to(dev), sn.to(dev), zero.to(dev), (1 - d).to(dev)
        drift = ops.mix3_per_sample(cond, x0, eps, dd, nd, zd)           # d*(cond - x0)
        noise = ops.mix3_per_sample(eps, x0, cond, sd, zd, zd)           # max_sigma*sqrt(n_t)*eps
        x_t = ops.mix3_per_sample(x0, cond, eps, od, dd, sd)             # (1-d)*x0 + d*cond + s*eps
        return t.to(dev), x_t, drift, eps, noise

    # ---- sampling ---------------------------------------------------------------------------------
    def _pred(self, net, a, b, t, names, text_encoder, image_context):
        out = net(a, b, t, names, text_encoder, image_context=image_context)
        return out[0] if isinstance(out, tuple) else out

    def predict(self, xa, x, cond, tdev, names, text_encoder, image_context):
        """(R_hat, eps_hat) of one denoising step.  The two networks are independent given the state, so on a GPU they
        are enqueued on two HIP streams: each net's small late-stage kernels (32x32 levels, token chains, kernel tails)
        overlap with the other net's work instead of leaving CUs idle."""
        if not (self.two_streams and xa.is_cuda):
            return (self._pred(self.drift_net, xa, cond, tdev, names, text_encoder, image_context),
                    self._pred(self.noise_net, xa, x, tdev, names, text_encoder, image_context))
        main = torch.cuda.current_stream()
        if self._streams is None:
            self._streams = (torch.cuda.Stream(), torch.cuda.Stream())
        s1, s2 = self._streams
        s1.wait_stream(main)
        s2.wait_stream(main)
        with torch.cuda.stream(s1):
            r_hat = self._pred(self.drift_net, xa, cond, tdev, names, text_encoder, image_context)
        with torch.cuda.stream(s2):
            e_hat = self._pred(self.noise_net, xa, x, tdev, names, text_encoder, image_context)
        main.wait_stream(s1)
        main.wait_stream(s2)
        r_hat.record_stream(main)
        e_hat.record_stream(main)
        return r_hat, e_hat

    @torch.no_grad()
    def reverse_ddpm(self, cond, names, text_encoder, reverse_type="std", optimize_type="inputRes", image_context=None, x_T=None,
                     noises=None, T_stop=0):
        """Iterative denoising from x_T = cond + max_sigma*z down to t=1.  `noises` (optional, [T, ...]) injects
        the per-step draws (parity runs; noises[i] is used at loop iteration i, t = T-i); x_T optional."""
        if optimize_type not in ("inputRes", "predict_noise", ""):
            raise NotImplementedError(f"optimize_type={optimize_type!r}: only the active 'inputRes' path of the reference "
                                      "(drift_noise_model.py:231-232) is in scope")
        cond = cond.contiguous()
        B = cond.shape[0]
        if x_T is None:
            x_T = ops.axpby(cond, self._randn_like(cond), 1.0, self.max_sigma)
        x = x_T.contiguous().clone()
        xa = ops.axpby(x, cond, 1.0, -1.0)
        x2, xa2 = torch.empty_like(x), torch.empty_like(x)
        tdev = torch.empty((B,), dtype=torch.float32, device=cond.device)
        nper = (x.numel() + 3) // 4
        for i, t in enumerate(range(self.T, T_stop, -1)):
            tdev.fill_(float(t))
            r_hat, e_hat = self.predict(xa, x, cond, tdev, names, text_encoder, image_context)
            z = None if noises is None else noises[i].contiguous()
            off = self._calls * nper
            self._calls += 1
            ops.drift_reverse_step(x, r_hat, e_hat, z, float(self._a[t]), float(self._b[t]), float(self._c[t]), cond=cond, seed=self.seed,
                                   offset=off, out=x2, xa_out=xa2)
            x, x2 = x2, x
            xa, xa2 = xa2, xa
        return x
