import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cgg_amd import ops
from cgg_amd.query_decoder import pack_bool_mask
dev = torch.device('cuda')
g = torch.Generator().manual_seed(0)
B, Q, H, E = 2, 100, 8, 256
for S in (1024, 4096, 16384):
    q = torch.randn(B, Q, E, generator=g).to(dev)
    k = torch.randn(B, S, E, generator=g).to(dev).bfloat16()
    vt = torch.randn(B, E, S, generator=g).to(dev).bfloat16()
    mask = (torch.rand(B, Q, S, generator=g) < 0.5).to(dev)
    bits = pack_bool_mask(mask)
    f = lambda: ops.masked_xattn_bf16(q, k, vt, bits, H)
    for _ in range(3): f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20): out = f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): gr.replay()
    e1.record(); torch.cuda.synchronize()
    print('S=%d: %.2f us (partial + combine), checksum %.6f' % (S, e0.elapsed_time(e1) * 1e3 / 200, out.float().abs().mean().item()))
