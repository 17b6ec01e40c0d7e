"""one denoising step (two nets on two streams + the fused update) repeated without intermediate synchronisation: are the nets' outputs
or the updated state what differs from run to run with the strip-form output layer?"""
import sys, torch
sys.path.insert(0, ".")
from instancediff_amd import ops, pipeline
from instancediff_amd.models.SDEs.driftSDE import driftSDE
from instancediff_amd.utils.synthetic import make_batch
from tests.test_sampling_gpu import make_scoremap_branch_visible
DEV = "cuda"
model, sde = pipeline.build(phase="test", device=torch.device(DEV), T=2, seed=0)
model.set_eval()
make_scoremap_branch_visible(model)
sde.hip_graph = False
b = make_batch(16, 256, seed=2024)
cond = b['input'].to(DEV).contiguous()
ctx = b['A_emb'].to(DEV).contiguous()
g = torch.Generator().manual_seed(2025)
x_T = (b['input'] + 0.4 * torch.randn(b['input'].shape, generator=g)).to(DEV)
noises = torch.randn((2,) + tuple(b['input'].shape), generator=g).to(DEV)
stash = []
orig = sde.predict
def spy(*a, **k):
    r, e = orig(*a, **k)
    stash.append((r, e))
    return r, e
sde.predict = spy
res = []
for i in range(5):
    st = driftSDE.Stepper(sde, x_T.clone(), cond, b['names'], model.text_encoder, ctx, noises=noises)
    with torch.no_grad():
        st._body()
        st._body()
    torch.cuda.synchronize()
    res.append((st.x.clone(), [(r.clone(), e.clone()) for r, e in stash]))
    stash.clear()
for i in range(1, 5):
    msg = []
    for s in range(2):
        for w, nm in ((0, "r_hat"), (1, "e_hat")):
            d = (res[i][1][s][w] - res[0][1][s][w]).abs()
            msg.append(f"step{s} {nm}: {int((d > 0).sum())} px")
    d = (res[i][0] - res[0][0]).abs()
    bad = (d > 0).nonzero()
    where = ""
    if bad.shape[0]:
        b0 = int(bad[0, 0])
        where = f" sample {b0} rows {sorted(set(bad[bad[:,0]==b0][:,2].tolist()))[:8]} cols {sorted(set(bad[bad[:,0]==b0][:,3].tolist()))[:20]}"
    print(f"run {i}: " + ", ".join(msg) + f" | final x: {bad.shape[0]} px differ" + where)
