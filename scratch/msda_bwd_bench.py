"""MSDeformAttn backward at configs[2] shapes (B images, encoder self-attention, grid-init offsets)."""
import sys, torch
sys.path.insert(0, '/root/repo')
import cgg_amd
from cgg_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device('cuda')
shapes = [(32, 32), (64, 64), (128, 128)]
H, D, P, L = 8, 32, 4, 3
starts, Nv = [], 0
for h, w in shapes:
    starts.append(Nv); Nv += h * w
g = torch.Generator().manual_seed(0)
value = torch.randn(B, Nv, H, D, generator=g).to(dev)
refs = []
for (h, w) in shapes:
    ys, xs = torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing='ij')
    refs.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
refp = torch.cat(refs, 0).to(dev)
wh = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32, device=dev)
spread = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
off = (torch.rand(B, Nv, H, L, P, 2, device=dev) - 0.5) * spread
loc = (refp[None, :, None, None, None, :] + off / wh[None, None, None, :, None, :]).contiguous()
aw = torch.softmax(torch.randn(B, Nv, H, L * P, device=dev), -1).view(B, Nv, H, L, P).contiguous()
go = torch.randn(B, Nv, H * D, device=dev)
ss = torch.tensor(shapes, dtype=torch.int64, device=dev)
st = torch.tensor(starts, dtype=torch.int64, device=dev)
for _ in range(2):
    ops.msda_backward(value, ss, st, loc, aw, go)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 5
e0.record()
for _ in range(n):
    ops.msda_backward(value, ss, st, loc, aw, go)
e1.record(); torch.cuda.synchronize()
print('msda_backward B=%d spread=%.1f px: %.3f ms (incl. 3 zero-fills)' % (B, spread, e0.elapsed_time(e1) / n))
