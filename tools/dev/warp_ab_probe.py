import sys, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pcfa_amd import hip_ops
def chk(t): return int(t.contiguous().view(torch.int32).to(torch.int64).sum().item())
out = []
for shape, scale in (((1, 32, 96, 320), 6.0), ((2, 24, 40, 72), 3.0), ((1, 7, 33, 70), 40.0), ((1, 16, 50, 35), 1.0)):
    g = torch.Generator().manual_seed(shape[1])
    B, C, H, W = shape
    x = torch.randn(*shape, generator=g).cuda().requires_grad_(True)
    go = torch.randn(*shape, generator=g).cuda()
    for kind in range(3):
        base = torch.nn.functional.interpolate(scale * torch.randn(B, 2, max(H // 8, 1), max(W // 8, 1), generator=g),
                                               size=(H, W), mode="bilinear", align_corners=False)
        f = (base, base + 0.3 * torch.randn(B, 2, H, W, generator=g), scale * torch.randn(B, 2, H, W, generator=g))[kind]
        flo = f.contiguous().cuda().requires_grad_(True)
        gx, gf = torch.autograd.grad(hip_ops.pwc_warp(x, flo, flow_scale=1.25), (x, flo), go)
        print(shape, kind, chk(gx), chk(gf), float(gx.abs().max()), int(torch.isnan(gx).sum()), int(torch.isnan(gf).sum()))
