"""f32 / bf16 MSDeformAttn forward at the bench geometry (B = 2, levels 32^2 / 64^2 / 128^2), init-like offsets."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cgg_amd
from cgg_amd import ops
dev = torch.device('cuda')
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
g = torch.Generator().manual_seed(0)
shapes = [(32, 32), (64, 64), (128, 128)]
B = 2
starts, N = [], 0
for h, w in shapes:
    starts.append(N); N += h * w
ref = torch.cat([torch.stack([(xs.flatten() + .5) / w, (ys.flatten() + .5) / h], -1) for h, w in shapes
                 for ys, xs in [torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing='ij')]]).to(dev)
th = torch.arange(8).float() * (2 * math.pi / 8)
d = torch.stack([th.cos(), th.sin()], -1); d = d / d.abs().max(-1, keepdim=True)[0]
grid = (d.view(8, 1, 1, 2) * torch.arange(1, 5).float().view(1, 1, 4, 1)).expand(8, 3, 4, 2)
rows = torch.cat([grid.reshape(1, 1, 192).expand(B, N, 192), torch.randn(B, N, 96, generator=g)], -1).contiguous().to(dev)
value = torch.randn(B, N, 8, 32, generator=g).to(dev)
v_hm = value.bfloat16().permute(0, 2, 1, 3).contiguous()
rows16 = rows.bfloat16()
a = ops.msda_forward_fused(value, shapes, starts, rows, ref, 4)
print('f32 %.1f us   bf16 hm %.1f us' % (timeit(lambda: ops.msda_forward_fused(value, shapes, starts, rows, ref, 4)),
                                         timeit(lambda: ops.msda_forward_fused_bf16(v_hm, shapes, starts, rows16, ref, 4, head_major=True))))
import hashlib
print('f32 output sha1', hashlib.sha1(a.cpu().numpy().tobytes()).hexdigest()[:16], 'T2D' if os.environ.get('CGG_MSDA_T2D') else 'strip')
# trained-like offsets: init grid + noise
rows2 = rows.clone(); rows2[..., :192] += torch.randn(B, N, 192, generator=g).to(dev) * 1.0
a2 = ops.msda_forward_fused(value, shapes, starts, rows2, ref, 4)
print('noisy offsets: f32 %.1f us sha1 %s' % (timeit(lambda: ops.msda_forward_fused(value, shapes, starts, rows2, ref, 4)),
                                             hashlib.sha1(a2.cpu().numpy().tobytes()).hexdigest()[:16]))
