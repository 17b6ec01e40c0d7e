"""r05 run19: where the grouped cross-attention's time goes -- the shipped kernel against builds without the S / P.V matrix products
(results meaningless, times informative).  One problem per level as in the c2 step (batch 16, 5 tokens x 4 heads)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from instancediff_amd import ops, _lib

dev = "cuda"
torch.manual_seed(0)
B, rows = 16, 20
levels = [(72, 256 * 256), (72, 128 * 128), (136, 64 * 64), (256, 32 * 32)]
lib = _lib.load()
groups = []
for Cm, N in levels:
    qf = torch.randn(B, 5, 4, Cm, device=dev) * 0.2
    mem = torch.randn(B, Cm, N, device=dev)
    groups.append((qf, mem))


def run(which):
    sel = [groups[i] for i in which]
    return ops.smm_xattn_grouped([g[0] for g in sel], [g[1] for g in sel], 0.25)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for which in ([0, 1, 2, 3], [0], [1], [2], [3]):
    print(f"levels {which}: {timed(lambda: run(which)):.1f} us (attention + merge launch)")
