#!/usr/bin/env python3
"""Repro driver for the open issue in DESIGN.md section 8: conv_wino4x, then fused-GroupNorm-tail launches, in one process.
usage: abort_repro.py MODE   (none | pack | conv | conv_sync)"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instancediff_amd import ops  # noqa: E402

mode = sys.argv[1]
dev = "cuda"
g = torch.Generator().manual_seed(1)
if mode != "none":
    x = torch.randn(17, 40, 64, 96, generator=g).to(dev)
    w = (torch.randn(64, 40, 3, 3, generator=g) / math.sqrt(360)).to(dev)
    wp = ops.pack_conv_weight(w)
    ops.attach_wino4x(wp, w)
    if mode in ("conv", "conv_sync"):
        for _ in range(3):
            y = ops.conv2d(x, wp, None, 3, 64, algo=ops.CONV_ALGO_WINOGRAD4X)
        if mode == "conv_sync":
            torch.cuda.synchronize()
            print("wino4x done", float(y.abs().mean()))
for B, C0, Cout, H, pro in ((16, 64, 64, 64, False), (3, 64, 64, 128, True), (5, 128, 256, 32, False)):
    x0 = torch.randn(B, C0, H, H, generator=g).to(dev)
    w = (torch.randn(Cout, C0, 3, 3, generator=g) / math.sqrt(9 * C0)).to(dev)
    bias = torch.randn(Cout, generator=g).to(dev)
    gamma, beta = torch.randn(Cout, generator=g).to(dev), torch.randn(Cout, generator=g).to(dev)
    kw = {}
    if pro:
        kw["pro"] = (torch.rand(B, C0, generator=g).to(dev) + 0.5, torch.randn(B, C0, generator=g).to(dev) * 0.1)
    wp2 = ops.pack_conv_weight(w)
    ticket = torch.zeros(4, dtype=torch.int32, device=dev)
    for rep in range(5):
        out, (a, b) = ops.conv2d(x0, wp2, bias, 3, Cout, gn=dict(groups=8, gamma=gamma, beta=beta, eps=1e-5, ticket=ticket), **kw)
    torch.cuda.synchronize()
    print("fused GN ok", B, C0, Cout, H, ticket.tolist(), float(a.abs().mean()))
print("done", mode)
