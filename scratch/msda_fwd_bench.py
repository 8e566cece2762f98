import sys, os, torch
sys.path.insert(0, '/root/repo')
import cgg_amd
from cgg_amd import ops
dev = torch.device('cuda')
B = 2
shapes = [(32, 32), (64, 64), (128, 128)]; starts = [0, 1024, 5120]; N = 21504
g = torch.Generator().manual_seed(0)
spread = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
raw = torch.randn(B, N, 288, generator=g); raw[..., :192] *= spread
ref = []
for h, w in shapes:
    ys, xs = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing='ij')
    ref.append(torch.stack([(xs.flatten() + .5) / w, (ys.flatten() + .5) / h], -1))
ref = torch.cat(ref).to(dev); raw16 = raw.to(dev).bfloat16()
v = torch.randn(B, N, 8, 32, generator=g).to(dev).bfloat16()
for mode in ('0', '1'):
    os.environ['CGG_MSDA_UNTILED'] = mode
    for _ in range(5): ops.msda_forward_fused_bf16(v, shapes, starts, raw16, ref, 4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.msda_forward_fused_bf16(v, shapes, starts, raw16, ref, 4)
    e1.record(); torch.cuda.synchronize()
    print('offset sigma %.1f px  untiled=%s: %.1f us' % (spread, mode, e0.elapsed_time(e1) / 50 * 1e3))
