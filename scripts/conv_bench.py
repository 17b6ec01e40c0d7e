#!/usr/bin/env python3
"""Micro-benchmark of the implicit-GEMM conv kernel on the UNet's own layer shapes (c2: B=16, 256^2).
Interleaved rounds in one process (cdna_hip_programming.md rule 24); prints TFLOP/s per shape/variant."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instancediff_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--algos", type=str, default="", help="comma list of conv algorithm ids to request for the 3x3 layers, timed interleaved "
                    "(3 = F(4x4,3x3) 16x32 items, 4 = half-patch form, 1 = F(2x2,3x3)); empty = the library's choice")
    ap.add_argument("--algos1", type=str, default="", help="the same for the 1x1 layers (0 = f32 matrix cores, 5 = bf16x3 split operands)")
    args = ap.parse_args()
    dev = "cuda"
    B = args.batch
    # (name, C0, C1, Cout, H, ks, mode, stats, pro)
    shapes = [
        ("L0 64->64 3x3 plain", 64, 0, 64, 256, 3, 0, False, False),
        ("L0 64->64 3x3 +stats", 64, 0, 64, 256, 3, 0, True, False),
        ("L0 64->64 3x3 +stats+pro", 64, 0, 64, 256, 3, 0, True, True),
        ("L0up 144->64 3x3 +stats", 64, 80, 64, 256, 3, 0, True, False),
        ("L1 64->64 3x3 +stats+pro", 64, 0, 64, 128, 3, 0, True, True),
        ("L2 128->128 3x3 +stats+pro", 128, 0, 128, 64, 3, 0, True, True),
        ("L3 256->256 3x3 +stats+pro", 256, 0, 256, 32, 3, 0, True, True),
        ("L3up 576->256 3x3 +stats", 256, 320, 256, 32, 3, 0, True, False),
        ("L2up 416->256 3x3 +stats", 256, 160, 256, 64, 3, 0, True, False),
        ("up 128->64 3x3 upsample", 128, 0, 64, 128, 3, 1, False, False),
        ("res 1x1 144->64", 64, 80, 64, 256, 1, 0, False, False),
        ("res 1x1 128->64", 64, 64, 64, 256, 1, 0, False, False),
        ("res 1x1 208->128", 128, 80, 128, 128, 1, 0, False, False),
        ("res 1x1 192->128", 128, 64, 128, 128, 1, 0, False, False),
        ("res 1x1 416->256", 256, 160, 256, 64, 1, 0, False, False),
        ("res 1x1 384->256", 256, 128, 256, 64, 1, 0, False, False),
        ("res 1x1 576->256", 256, 320, 256, 32, 1, 0, False, False),
        ("res 1x1 512->256", 256, 256, 256, 32, 1, 0, False, False),
        ("qkv 1x1 256->768", 256, 0, 768, 32, 1, 0, False, False),
        ("proj 1x1 256->256", 256, 0, 256, 32, 1, 0, False, False),
        ("mem 1x1 64->256", 64, 0, 256, 256, 1, 0, False, False),
        ("down 1x1 unshuffle 64->64", 64, 0, 64, 256, 1, 2, False, False),
        ("down 1x1 unshuffle 128->256 @64", 128, 0, 256, 64, 1, 2, False, False),
        ("down 1x1 unshuffle 64->128 @128", 64, 0, 128, 128, 1, 2, False, False),
        ("final 3x3 64->5", 64, 0, 5, 256, 3, 0, False, False),
        ("init 7x7 2->64", 1, 1, 64, 256, 7, 0, False, False),
    ]
    cases = []
    if args.only:
        shapes = [s for s in shapes if args.only in s[0]]
    for name, C0, C1, Co, H, ks, mode, stats, pro in shapes:
        x0 = torch.randn(B, C0, H, H, device=dev)
        x1 = torch.randn(B, C1, H, H, device=dev) if C1 else None
        cin = (C0 * 4 if mode == 2 else C0) + C1
        wraw = torch.randn(Co, cin, ks, ks, device=dev) / (cin * ks * ks) ** 0.5
        w = ops.pack_conv_weight(wraw)
        b = torch.randn(Co, device=dev)
        p = (torch.rand(B, C0, device=dev) + 0.5, torch.randn(B, C0, device=dev) * 0.1) if pro else None
        Ho = H * 2 if mode == 1 else (H // 2 if mode == 2 else H)
        out = torch.empty(B, Co, Ho, Ho, device=dev)
        fl = 2.0 * cin * Co * ks * ks * Ho * Ho * B
        kw = dict(src0=x0, wpk=w, bias=b, ks=ks, Cout=Co, src1=x1, mode=mode, pro=p, out=out, want_stats=stats)
        if args.algos and ks == 3:
            for al in args.algos.split(","):
                cases.append((f"{name} [algo {al}]", dict(kw, algo=int(al)), fl))
        elif args.algos1 and ks == 1 and mode in (0, 2):
            for al in args.algos1.split(","):
                cases.append((f"{name} [algo {al}]", dict(kw, algo=int(al)), fl))
        else:
            cases.append((name, kw, fl))
    times = {c[0]: [] for c in cases}
    for r in range(args.rounds + 1):
        for name, kw, fl in cases:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                try:
                    ops.conv2d(**kw)
                except Exception as e:  # a requested kernel that does not tile this shape
                    times[name].append(float("nan"))
                    break
            e1.record()
            torch.cuda.synchronize()
            if r > 0:
                times[name].append(e0.elapsed_time(e1) / args.iters)
    for name, kw, fl in cases:
        t = sorted(v for v in times[name] if v == v)
        if not t:
            print(f"{name:44s} not applicable")
            continue
        med = t[len(t) // 2]
        print(f"{name:44s} {med * 1e3:9.1f} us  {fl / med / 1e9:7.1f} TFLOP/s  (min {t[0] * 1e3:.1f} us)")


if __name__ == "__main__":
    main()
