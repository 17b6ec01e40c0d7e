#!/usr/bin/env python3
"""Where a F(4x4,3x3) kernel's output differs from the direct kernel: per shape, the error map reduced over samples / channels.
usage: w4_diag.py algo B C0 Cout H W [pro]"""
import math
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from instancediff_amd import ops  # noqa: E402

algo, B, C0, Cout, H, W = [int(v) for v in sys.argv[1:7]]
pro = len(sys.argv) > 7
g = torch.Generator().manual_seed(3)
x = torch.randn(B, C0, H, W, generator=g).cuda()
w = (torch.randn(Cout, C0, 3, 3, generator=g) / math.sqrt(9 * C0)).cuda()
b = torch.randn(Cout, generator=g).cuda()
kw = {}
if pro:
    kw["pro"] = (torch.rand(B, C0, generator=g).cuda() + 0.5, torch.randn(B, C0, generator=g).cuda() * 0.1)
wp = ops.pack_conv_weight(w)
for rep in range(2):
    o1, s1 = ops.conv2d(x, wp, b, 3, Cout, want_stats=True, algo=algo, **kw)
    o0, s0 = ops.conv2d(x, wp, b, 3, Cout, want_stats=True, algo=ops.CONV_ALGO_DIRECT, **kw)
    torch.cuda.synchronize()
    d = (o1 - o0).abs()
    print(f"algo {algo} B={B} C0={C0} Cout={Cout} {H}x{W} pro={pro} rep {rep}: max err {float(d.max()):.3e}, stats err {float((s1 - s0).abs().max()):.3e}")
    if float(d.max()) > 1e-3:
        bad = d > 1e-3
        print("  bad fraction", float(bad.float().mean()))
        print("  by sample   ", [round(float(v), 3) for v in bad.float().mean(dim=(1, 2, 3))])
        print("  by 16-ch blk", [round(float(v), 3) for v in bad.float().mean(dim=(0, 2, 3)).reshape(-1, 16).mean(1)])
        rows = bad.float().mean(dim=(0, 1, 3))
        cols = bad.float().mean(dim=(0, 1, 2))
        print("  by row      ", "".join("#" if v > 0.5 else ("+" if v > 0 else ".") for v in rows))
        print("  by col      ", "".join("#" if v > 0.5 else ("+" if v > 0 else ".") for v in cols))
        idx = bad.nonzero()[:12]
        sflat = s0.reshape(-1)
        for bb, cc, yy, xx in idx.tolist():
            got, want = float(o1[bb, cc, yy, xx]), float(o0[bb, cc, yy, xx])
            near = (sflat - got).abs()
            k = int(near.argmin())
            tag = f"= stats[{k}] (b,tile,c,which = {k // (s0.shape[1] * s0.shape[2] * 2)},{(k // (s0.shape[2] * 2)) % s0.shape[1]},{(k // 2) % s0.shape[2]},{k % 2})" if float(near[k]) < 1e-6 * max(1.0, abs(got)) else ""
            same = (o0 == got).nonzero()[:2].tolist()
            print(f"    (b{bb} c{cc} y{yy} x{xx}) got {got:.6f} want {want:.6f} {tag} equals o0 at {same}")
