import sys, os, torch
sys.path.insert(0, os.getcwd())
from instancediff_amd import ops
dev = "cuda"
for (B, C, H, Cm) in ((16, 64, 256, 72), (16, 64, 128, 72), (16, 128, 64, 136)):
    feat = torch.randn(B, C + 16, H, H, device=dev)[:, :C]
    g1, b1 = torch.randn(C, device=dev), torch.randn(C, device=dev)
    lin = torch.nn.Linear(C, 256).to(dev)
    gram, hvec, evar = ops.memory_variance_form(lin.weight, lin.bias)
    for _ in range(3):
        ops.smm_memproj_compact(feat, g1, b1, gram, hvec, evar, Cm)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.smm_memproj_compact(feat, g1, b1, gram, hvec, evar, Cm)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    bytes_ = B * H * H * 4 * (C + Cm)
    print(f"memproj_compact B={B} C={C} {H}x{H} Cm={Cm}: {us:.1f} us  {bytes_ / us / 1e6:.2f} TB/s")
