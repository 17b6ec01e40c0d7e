#!/bin/bash
# r05 run 11: which r05 change broke the bitwise batch invariance of the 256^2 chain (collect A2)?  B=16 vs B=5 on the same samples, with the
# strip form of conv3x3_select on / off and the grouped ScoreMapModule launches on / off
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/inv.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
from instancediff_amd import pipeline
from instancediff_amd.utils.synthetic import make_batch
from tests.test_sampling_gpu import make_scoremap_branch_visible
DEV = "cuda"
T, H = 2, 256
model, sde = pipeline.build(phase="test", device=torch.device(DEV), T=T, seed=0)
model.set_eval()
make_scoremap_branch_visible(model)
b16 = make_batch(16, H, seed=2024)
g = torch.Generator().manual_seed(2025)
x_T = b16['input'] + 0.4 * torch.randn(b16['input'].shape, generator=g)
noises = torch.randn((T,) + tuple(b16['input'].shape), generator=g)
def chain(batch, x, n):
    model.feed_data(batch)
    model.test(x_T=x.to(DEV), noises=n.to(DEV))
    return torch.from_numpy(model.get_visuals()).clone()
sl = lambda b, n: {k: v[:n] for k, v in b.items()}
o16 = chain(b16, x_T, noises)
o16b = chain(b16, x_T, noises)
o5 = chain(sl(b16, 5), x_T[:5], noises[:, :5].contiguous())
d = (o16[:5] - o5).abs()
print("B16 vs B16 again:", float((o16 - o16b).abs().max()), "| B16[:5] vs B5:", float(d.max()), "per sample:", [float(d[i].max()) for i in range(5)],
      "| pixels differing:", int((d > 0).sum()))
PY
for cfg in "" "IDIFF_SELECT_STRIPS=0" "IDIFF_GROUPED_SMM=0" "IDIFF_HIP_GRAPH=0"; do
  echo "== [$cfg]"; env $cfg python3 /tmp/inv.py 2>&1 | tail -1
done
