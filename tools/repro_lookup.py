import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcfa_amd import hip_ops
def say(*a):
    print(*a, flush=True)
for (B, D, H, W) in ((1, 64, 16, 20), (2, 32, 17, 21), (1, 256, 55, 128)):
    g = torch.Generator().manual_seed(0)
    f1 = torch.randn(B, D, H, W, generator=g).cuda().requires_grad_(True)
    f2 = torch.randn(B, D, H, W, generator=g).cuda().requires_grad_(True)
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    coords = (torch.stack([xs, ys], 0).float()[None].repeat(B, 1, 1, 1) + 3 * torch.randn(B, 2, H, W, generator=g)).cuda()
    say("shape", B, D, H, W)
    blk = hip_ops.CorrBlock(f1, f2); torch.cuda.synchronize(); say("  build ok")
    out = blk(coords); torch.cuda.synchronize(); say("  lookup fwd ok", float(out.abs().sum()))
    out.sum().backward(); torch.cuda.synchronize(); say("  backward ok", float(f1.grad.abs().sum()))
