import torch, sys, math, numpy as np
sys.path.insert(0, '.')
from instancediff_amd import ops
from oracle import sde_ref
gd = np.load('tests/golden/irsde_golden.npz')
sde = sde_ref.IRSDERef(0.4, T=100, sample_T=50)
mu = torch.from_numpy(gd["t64/mu"]); x = torch.from_numpy(gd["t64/xT"]); noises = torch.from_numpy(gd["t64/noises"])
sde.set_mu(mu)
g = torch.Generator().manual_seed(10)
t = 50
npred = torch.randn(x.shape, generator=g)
score = sde.get_score_from_noise(npred, t)
kw = dict(theta=float(sde.thetas[t]), sigma=float(sde.sigmas[t]), sigma_bar=float(sde.sigma_bars[t]), dt=float(sde.dt), sqrt_dt=math.sqrt(float(sde.dt)))
for mode, ref in [(0, sde.reverse_sde_step(x, score, t, noises[0])), (1, sde.reverse_sde_step_mean(x, score, t)), (2, sde.reverse_ode_step(x, score, t))]:
    out = ops.irsde_reverse_step(x.cuda(), mu.cuda(), npred.cuda(), noises[0].cuda() if mode == 0 else None, mode=mode, **kw).cpu()
    d = (out - ref)
    print(mode, "ndiff", int((d != 0).sum()), "of", d.numel(), float(d.abs().max()))
# isolate: dispersion only: x=0, mu=0, npred=0
z = noises[0]
zero = torch.zeros_like(x)
out = ops.irsde_reverse_step(zero.cuda(), zero.cuda(), zero.cuda(), z.cuda(), mode=0, **kw).cpu()
ref = zero - sde.sde_reverse_drift(zero, zero, t) - sde.dispersion(zero, t, z)
print("disp only ndiff", int((out != ref).sum()))
ref2 = -(sde.sigmas[t] * (z * np.float32(math.sqrt(float(sde.dt)))))
print("disp vs f32 scalar", int((out != ref2).sum()), "oracle vs f32", int((ref != ref2).sum()))
print(repr(math.sqrt(float(sde.dt))), repr(float(np.float32(math.sqrt(float(sde.dt))))))
