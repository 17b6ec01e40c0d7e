"""r05 run18: smm_memproj (weight block prefetch) and linear_t_grouped (weights requested before the rows are staged): times + fp64 check"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from instancediff_amd import ops

dev = "cuda"
torch.manual_seed(0)


def timed(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (B, C, H) in ((16, 256, 32), (16, 256, 64), (16, 128, 128), (32, 128, 128), (2, 72, 24)):
    feat = torch.randn(B, C, H, H, device=dev)
    g1, b1 = torch.randn(C, device=dev), torch.randn(C, device=dev)
    lin = torch.nn.Linear(C, 256).to(dev)
    g2, b2 = torch.randn(256, device=dev), torch.randn(256, device=dev)
    wpk = lin.weight.detach().t().contiguous()
    out = ops.smm_memproj(feat, g1, b1, wpk, lin.bias.detach(), g2, b2)
    x = feat.double().flatten(2).transpose(1, 2)
    ref = torch.nn.functional.layer_norm(x, (C,), g1.double(), b1.double(), 1e-5) @ lin.weight.double().t() + lin.bias.double()
    ref = torch.nn.functional.layer_norm(ref, (256,), g2.double(), b2.double(), 1e-5).transpose(1, 2)
    err = (out.double() - ref).abs().max().item()
    out2 = ops.smm_memproj(feat, g1, b1, wpk, lin.bias.detach(), g2, b2)
    us = timed(lambda: ops.smm_memproj(feat, g1, b1, wpk, lin.bias.detach(), g2, b2))
    print(f"smm_memproj B={B} C={C} {H}x{H}: {us:.1f} us  max|err| vs fp64 {err:.2e}  run-to-run equal {bool((out == out2).all())}")

for (R, K, N, ngroups, ln) in ((80, 256, 256, 4, False), (80, 256, 768, 4, True), (80, 256, 1024, 4, True), (80, 1024, 256, 4, False), (80, 512, 256, 2, False),
                               (37, 100, 52, 3, True)):
    gs = []
    for _ in range(ngroups):
        x = torch.randn(R, K, device=dev)
        wT = torch.randn(K, N, device=dev) / K ** 0.5
        bias = torch.randn(N, device=dev)
        g = dict(x=x, wT=wT, bias=bias)
        if ln:
            g["ln"] = (torch.randn(K, device=dev), torch.randn(K, device=dev), 1e-5)
        gs.append(g)
    outs = ops.linear_t_grouped(gs)
    err = 0.0
    for g, o in zip(gs, outs):
        x = g["x"].double()
        if ln:
            x = torch.nn.functional.layer_norm(x, (K,), g["ln"][0].double(), g["ln"][1].double(), 1e-5)
        err = max(err, (o.double() - (x @ g["wT"].double() + g["bias"].double())).abs().max().item())
    singles = [ops.linear_t(g["x"], g["wT"], bias=g["bias"], ln=g.get("ln")) for g in gs]
    same = all(bool((a == b).all()) for a, b in zip(outs, singles))
    us = timed(lambda: ops.linear_t_grouped(gs))
    print(f"linear_t_grouped {ngroups} x [{R}x{K}]x[{K}x{N}] ln={ln}: {us:.1f} us  max|err| vs fp64 {err:.2e}  grouped == single {same}")
