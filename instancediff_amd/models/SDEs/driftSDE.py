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
        self._calls = 0  # draws made
        self._off = 0    # Philox counters consumed: draws of any mix of sizes use disjoint counter ranges
        self.two_streams = bool(int(os.environ.get("IDIFF_TWO_STREAMS", "1")))
        self.hip_graph = bool(int(os.environ.get("IDIFF_HIP_GRAPH", "1")))
        self._streams = None

    def set_gpu(self, device):
        self.device = device
        self.drift_schedule = self._h_drift.to(device)
        self.noise_schedule = self._h_noise.to(device)

    def set_seed(self, seed):
        self.seed = int(seed)
        self._calls = 0
        self._off = 0

    def _randn_like(self, x):
        off = self._off
        self._calls += 1
        self._off += (x.numel() + 3) // 4
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
        from ..modules import MSM_degEmb_Unet as _U
        with torch.cuda.stream(s1):
            r_hat = self._pred(self.drift_net, xa, cond, tdev, names, text_encoder, image_context)
        with torch.cuda.stream(s2):
            e_hat = self._pred(self.noise_net, xa, x, tdev, names, text_encoder, image_context)
        main.wait_stream(s1)
        main.wait_stream(s2)
        if _U.SMM_SIDE:  # (experiment) a net's decoder ran on its side stream, which only the origin may join (MSM_degEmb_Unet.SMM_SIDE)
            for net in (self.drift_net, self.noise_net):
                if getattr(net, "_side_stream", None) is not None:
                    main.wait_stream(net._side_stream)
        r_hat.record_stream(main)
        e_hat.record_stream(main)
        return r_hat, e_hat

    # ---- one denoising step as a replayable unit ----------------------------------------------------
    class Stepper:
        """The body of the reverse loop with every per-step scalar in device memory (timestep vector, (a_t, b_t, c_t) tables,
        Philox call count, step index), so the same launches serve every t -- eagerly, or as ONE captured HIP graph that is
        replayed per step (`IDIFF_HIP_GRAPH=0` disables the capture).  The graph holds the two UNet forwards on their two
        streams, the fused update and the state advance; the host does not touch the loop between replays."""

        def __init__(self, sde, x, cond, names, text_encoder, image_context, noises=None, t_start=None, t_stop=0):
            self.sde, self.names, self.text_encoder, self.ctx = sde, names, text_encoder, image_context
            dev = x.device
            self.x, self.cond = x, cond
            self.xa = ops.axpby(x, cond, 1.0, -1.0)
            self.T, self.t_stop = sde.T, int(t_stop)
            t0 = sde.T if t_start is None else int(t_start)
            self.tdev = torch.full((x.shape[0],), float(t0), dtype=torch.float32, device=dev)
            self.coef = torch.stack([sde._a, sde._b, sde._c]).to(device=dev, dtype=torch.float32).contiguous()
            self.state = torch.tensor([t0, 0, 0], dtype=torch.int32, device=dev)  # {t, draws of this run, step index}
            self.noises = None if noises is None else noises.contiguous()
            self.nper = (x.numel() + 3) // 4
            self.off_base = sde._off  # this run's draws start where the stream's earlier ones ended
            self.graph = None
            self.steps_done = 0

        def _body(self):
            sde = self.sde
            r_hat, e_hat = sde.predict(self.xa, self.x, self.cond, self.tdev, self.names, self.text_encoder, self.ctx)
            ops.drift_reverse_step_dev(self.x, r_hat, e_hat, self.noises, self.cond, self.xa, self.coef, self.state, sde.seed, self.nper, self.off_base)
            ops.step_state_advance(self.state, self.tdev, self.T, self.t_stop)

        def _warm_step(self):
            """one step eagerly on the side stream: fills every weight / text cache outside the graph's memory pool.  It IS a
            denoising step (x, xa and the device state {t, Philox count, step index} advance), whatever happens to the capture."""
            main = torch.cuda.current_stream()
            self.stream = torch.cuda.Stream()
            self.stream.wait_stream(main)
            with torch.cuda.stream(self.stream):
                self._body()
            main.wait_stream(self.stream)

        def _capture(self):
            """capture the next step as a HIP graph (enqueues nothing that executes)"""
            main = torch.cuda.current_stream()
            self.stream.wait_stream(main)
            with torch.cuda.stream(self.stream):
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                # thread-local capture mode: a polling thread of the process (e.g. the RCCL watchdog of a multi-GPU job) must
                # not invalidate the capture
                with torch.cuda.graph(g, stream=self.stream, capture_error_mode="thread_local"):
                    self._body()
            main.wait_stream(self.stream)
            return g

        @torch.no_grad()
        def prepare(self):
            """Warm step + graph capture; returns the number of denoising steps it executed (1, counted in steps_done, also when
            the capture then fails and the loop stays eager); 0 when capture is off or already attempted."""
            sde = self.sde
            if not (sde.hip_graph and self.x.is_cuda and self.graph is None):
                return 0
            self._warm_step()
            self.steps_done += 1
            sde._calls += 1
            sde._off += self.nper
            try:
                self.graph = self._capture()
            except Exception as e:  # stay on the eager HIP path (same kernels), say why once
                self.graph = False
                print(f"[instancediff_amd] HIP graph capture unavailable ({e!r}); running the step eagerly")
            return 1

        @torch.no_grad()
        def run(self, nsteps):
            sde = self.sde
            nsteps -= self.prepare() if nsteps >= 3 else 0
            if self.graph:
                main = torch.cuda.current_stream()
                self.stream.wait_stream(main)
                with torch.cuda.stream(self.stream):
                    for _ in range(nsteps):
                        self.graph.replay()
                main.wait_stream(self.stream)
            else:
                for _ in range(nsteps):
                    self._body()
            self.steps_done += nsteps
            sde._calls += nsteps
            sde._off += nsteps * self.nper
            return self.x

        @property
        def mode(self):
            """'graph' when the loop replays a captured HIP graph, 'eager' otherwise (reported by bench.py)"""
            return "graph" if self.graph else "eager"

    @torch.no_grad()
    def reverse_ddpm(self, cond, names, text_encoder, reverse_type="std", optimize_type="inputRes", image_context=None, x_T=None,
                     noises=None, T_stop=0):
        """Iterative denoising from x_T = cond + max_sigma*z down to t=1.  `noises` (optional, [T, ...]) injects
        the per-step draws (parity runs; noises[i] is used at loop iteration i, t = T-i); x_T optional."""
        if optimize_type not in ("inputRes", "predict_noise", ""):
            raise NotImplementedError(f"optimize_type={optimize_type!r}: only the active 'inputRes' path of the reference "
                                      "(drift_noise_model.py:231-232) is in scope")
        if "std" not in str(reverse_type):
            # optimize_target (drift_noise_model.py:68,581-604): 'std*' nets predict LQ-GT and the standard noise, which is what
            # the update consumes; 'scaled*' nets predict d_t*(LQ-GT) and s_t*eps and would need rescaling -- not silently ignored
            raise NotImplementedError(f"reverse_type={reverse_type!r}: only the 'std' prediction targets of config.yml:144 are in scope")
        cond = cond.contiguous()
        B = cond.shape[0]
        if x_T is None:
            x_T = ops.axpby(cond, self._randn_like(cond), 1.0, self.max_sigma)
        x = x_T.contiguous().clone()
        stepper = driftSDE.Stepper(self, x, cond, names, text_encoder, image_context, noises=noises, t_stop=T_stop)
        out = stepper.run(self.T - T_stop)
        self.last_mode = stepper.mode  # 'graph' | 'eager': how the loop of this call ran
        return out
