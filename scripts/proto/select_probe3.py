"""two nets on two streams (driftSDE.predict), repeated: where do the strip-form output layer's run-to-run differences sit?"""
import sys, torch
sys.path.insert(0, ".")
from instancediff_amd import ops, pipeline
from instancediff_amd.utils.synthetic import make_batch
from tests.test_sampling_gpu import make_scoremap_branch_visible
DEV = "cuda"
model, sde = pipeline.build(phase="test", device=torch.device(DEV), T=2, seed=0)
model.set_eval()
make_scoremap_branch_visible(model)
b = make_batch(16, 256, seed=2024)
cond = b['input'].to(DEV).contiguous()
x = (cond + 0.1).contiguous()
xa = ops.axpby(x, cond, 1.0, -1.0)
ctx = b['A_emb'].to(DEV).contiguous()
t = torch.full((16,), 2.0, device=DEV)
outs = []
with torch.no_grad():
    for i in range(6):
        r, e = sde.predict(xa, x, cond, t, b['names'], model.text_encoder, ctx)
        torch.cuda.synchronize()
        outs.append((r.clone(), e.clone()))
for i in range(1, 6):
    for which, name in ((0, "drift"), (1, "noise")):
        d = (outs[i][which] - outs[0][which]).abs()
        bad = (d > 0).nonzero()
        if bad.shape[0]:
            bs = sorted(set(bad[:, 0].tolist()))
            b0 = bs[0]
            rows = sorted(set(bad[bad[:, 0] == b0][:, 2].tolist()))
            cols = sorted(set(bad[bad[:, 0] == b0][:, 3].tolist()))
            print(f"run {i} {name}: {bad.shape[0]} px differ, max {float(d.max()):.3e}, samples {bs}; sample {b0}: rows {rows[:10]} cols {cols[:24]}")
        else:
            print(f"run {i} {name}: identical")
