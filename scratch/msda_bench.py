import sys, torch
sys.path.insert(0, '.')
import cgg_amd
from cgg_amd import ops
dev = 'cuda'
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
B = 2; shapes = [(32, 32), (64, 64), (128, 128)]
starts = [0, 1024, 1024 + 4096]; N = 21504
g = torch.Generator().manual_seed(0)
raw = torch.randn(B, N, 288, generator=g)
raw[..., :192] *= 2.0   # offsets of a few pixels
ref = []
for h, w in shapes:
    ys, xs = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing='ij')
    ref.append(torch.stack([(xs.flatten() + .5) / w, (ys.flatten() + .5) / h], -1))
ref = torch.cat(ref).to(dev); raw = raw.to(dev)
for dt in (torch.float32, torch.bfloat16):
    v = torch.randn(B, N, 8, 32, generator=g).to(dev).to(dt)
    t = bench(lambda: ops.msda_forward_fused(v, shapes, starts, raw, ref, 4))
    byts = B * N * (256 * v.element_size() + 288 * 4 + 256 * 4)
    print(dt, '%.1f us' % t, '%.0f GB/s algorithmic' % (byts / t / 1e3))
