import sys, torch
sys.path.insert(0, ".")
from instancediff_amd import ops
torch.manual_seed(0)
B, C, H, W, K = 16, 64, 256, 256, 5
x = torch.randn(B, C, H, W, device="cuda")
w = torch.randn(K, C, 3, 3, device="cuda") * 0.05
bias = torch.randn(K, device="cuda")
idx = torch.tensor([i % K for i in range(B)], dtype=torch.int32, device="cuda")
outs = [ops.conv3x3_select(x, w, bias, idx).clone() for _ in range(6)]
torch.cuda.synchronize()
ref = torch.nn.functional.conv2d(x.double(), w.double(), bias.double(), padding=1)[torch.arange(B), idx.long()][:, None]
for i, o in enumerate(outs):
    d = (o.double() - ref).abs()
    bad = (d > 1e-4).nonzero()
    print("run", i, "max err vs fp64", float(d.max()), "bad pixels", bad.shape[0], "equal to run 0:", bool(torch.equal(o, outs[0])), bad[:6].tolist())
# interleaved with another kernel writing x-sized memory right before (a producer), as in the UNet
y = torch.empty_like(x)
for i in range(3):
    ops.axpby(x, x, 1.0, 0.0, out=y)
    o = ops.conv3x3_select(y, w, bias, idx)
    d = (o.double() - ref).abs()
    print("after producer", i, "max err", float(d.max()), "bad", int((d > 1e-4).sum()))
