import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cgg_amd import ops
dev = 'cuda'
g = torch.Generator().manual_seed(0)
B = 2
shapes = [(32, 32), (64, 64), (128, 128)]; starts = [0, 1024, 5120]; N = 21504
raw = torch.randn(B, N, 288, generator=g); raw[..., :192] *= 2.0
ref = []
for h, w in shapes:
    ys, xs = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing='ij')
    ref.append(torch.stack([(xs.flatten() + .5) / w, (ys.flatten() + .5) / h], -1))
ref = torch.cat(ref).to(dev); raw16 = raw.to(dev).bfloat16()
v = torch.randn(B, N, 8, 32, generator=g).to(dev).bfloat16()
f = lambda: ops.msda_forward_fused_bf16(v, shapes, starts, raw16, ref, 4)
for _ in range(3): out = f()
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for _ in range(20): out = f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): gr.replay()
e1.record(); torch.cuda.synchronize()
print('msda bf16 stream: %.2f us  checksum %.6f' % (e0.elapsed_time(e1) * 1e3 / 100, out.float().abs().mean().item()))
