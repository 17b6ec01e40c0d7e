import torch, sys
sys.path.insert(0, '.')
from instancediff_amd.models.modules import MSM_degEmb_Unet as P
from oracle import unet_ref as R
torch.manual_seed(0)
for (B, C, H, M) in [(2, 64, 32, 4), (1, 64, 32, 4), (1, 256, 8, 4), (2, 128, 16, 2)]:
    ca = P.CrossAttention(C, 512, 4)
    with torch.no_grad():
        ca.norm.weight.normal_(1, 0.1); ca.norm.bias.normal_(0, 0.1)
    rc = R.CrossAttention(C, 512, 4); rc.load_state_dict(ca.state_dict())
    x = torch.randn(B, C, H, H); ctx = torch.randn(B, M, 512)
    with torch.no_grad():
        ref = x + rc(x, ctx)
        ca = ca.cuda()
        out = ca.run(x.cuda(), ctx.cuda())
        xn = P.ops.chan_layernorm(x.cuda(), ca.norm.weight, ca.norm.bias)
        e0 = (xn.cpu() - rc.norm(x)).abs().max()
        q = P.ops.conv2d(xn, P.packed(ca.q_proj), None, 1, C)
        e1 = (q.cpu() - rc.q_proj(rc.norm(x))).abs().max()
    print(B, C, H, M, "err", float((out.cpu() - ref).abs().max()), float(e0), float(e1))
