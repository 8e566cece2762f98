import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cgg_amd
from cgg_amd import ops
dev = torch.device('cuda')
shapes = [(8, 8), (16, 16), (32, 32)]
starts, Nv = [0, 64, 320], 1344
g = torch.Generator().manual_seed(60)
B, H, D, L, P = 2, 8, 32, 3, 4
value = torch.randn(B, Nv, H, D, generator=g).to(dev)
loc = torch.rand(B, Nv, H, L, P, 2, generator=g).to(dev)
aw = torch.softmax(torch.randn(B, Nv, H, L * P, generator=g), -1).view(B, Nv, H, L, P).to(dev)
go = torch.randn(B, Nv, H * D, generator=g).to(dev)
ss = torch.tensor(shapes, dtype=torch.int64, device=dev); st = torch.tensor(starts, dtype=torch.int64, device=dev)
gv, gl, ga = ops.msda_backward(value, ss, st, loc, aw, go)
gv2, gl2, ga2 = ops.msda_backward_hostlevels(value, shapes, starts, loc, aw, go)
for n, a, b in (('gv', gv, gv2), ('gl', gl, gl2), ('ga', ga, ga2)):
    d = (a - b).abs()
    print(n, 'max diff', d.max().item(), 'n diff', int((d > 0).sum()), 'of', d.numel(), 'nan', int(torch.isnan(a).sum()), int(torch.isnan(b).sum()))
    if (d > 0).any():
        idx = torch.nonzero(d > 0)[:5]
        print(idx.tolist(), [a[tuple(i)].item() for i in idx], [b[tuple(i)].item() for i in idx])
