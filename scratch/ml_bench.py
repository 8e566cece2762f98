import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cgg_amd import ops
dev = 'cuda'
B, Q, C, H, W = 2, 100, 256, 256, 256
g = torch.Generator().manual_seed(0)
E = torch.randn(B, Q, C, generator=g).to(dev)
F_ = torch.randn(B, C, H, W, generator=g).to(dev)
def timed(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n): out = f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n), out
for split in (False, True):
    packed = ops.pack_mask_feature(F_, 1, split=split)
    us, out = timed(lambda: ops.mask_logits(E, packed, want_logits=True)[0])
    print('split' if split else 'bf16 ', 'full: %.2f us' % us, 'checksum %.4f' % out.abs().mean().item())
pooled = ops.pack_mask_feature(F_, 8, split=False)
us, out = timed(lambda: ops.mask_logits(E, pooled, want_logits=False, want_bits=True)[1])
print('bits 32x32: %.2f us' % us, int(out.sum()))
