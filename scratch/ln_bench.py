import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cgg_amd import ops
dev = 'cuda'
g = torch.Generator().manual_seed(0)
B, N, C = 2, 21504, 256
a = torch.randn(B, N, C, generator=g).bfloat16().to(dev); b = torch.randn(B, N, C, generator=g).bfloat16().to(dev)
gamma = torch.randn(C, generator=g).to(dev); beta = torch.randn(C, generator=g).to(dev); pos = torch.randn(N, C, generator=g).to(dev)
def timed(f, n=30):
    for _ in range(3): f()
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
us1, o1 = timed(lambda: ops.add_layernorm_stream(a, b, gamma, beta, 1e-5, want_f32=False))
us2, o2 = timed(lambda: ops.add_layernorm_stream(a, b, gamma, beta, 1e-5, pos=pos, want_f32=False, want_pos=True))
ref = torch.nn.functional.layer_norm(a.float() + b.float(), (C,), gamma, beta, 1e-5)
print('LN1 (y16): %.2f us   LN2 (y16 + yp16): %.2f us   max err %.4f / %.4f' % (us1, us2, (o1[1].float() - ref).abs().max().item(), (o2[2].float() - (ref + pos)).abs().max().item()))
