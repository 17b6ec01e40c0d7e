"""one problem of the grouped cross-attention (the 65 536-key level of c2: batch 16, 20 query rows, Cm = 72), a few launches -- for --pmc passes"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from instancediff_amd import ops
torch.manual_seed(0)
dev = "cuda"
B, Cm, N = 16, 72, int(os.environ.get("XA_N", 65536))
qf = torch.randn(B, 5, 4, Cm, device=dev) * 0.2
mem = torch.randn(B, Cm, N, device=dev)
for _ in range(4):
    ops.smm_xattn_grouped([qf], [mem], 0.25)
torch.cuda.synchronize()
