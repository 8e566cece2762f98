"""decoder tail: one launch vs the 5-launch chain (hipGraph replay timing)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from cgg_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
M, C, Q, nsum = 200, 256, 100, 8
planes = (torch.randn(nsum, M, C, generator=g) * 0.5).to(dev)
pos = torch.randn(Q, C, generator=g).to(dev)
na = (torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev), 1e-5)
nb = (torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev), 1e-5)
ws = [(torch.randn(C, C, generator=g) / 16).to(dev) for _ in range(4)]
bs = [torch.randn(C, generator=g).to(dev) for _ in range(4)]
pk = [ops.pack_linear_weight(w) for w in ws]
mlp = (pk[0], bs[0], pk[1], bs[1], pk[2], bs[2])
def fused():
    return ops.decoder_tail(planes, na, pos, nb, mlp, (pk[3], bs[3]))
def chain():
    y0, yp0, z0 = ops.layernorm_chain(planes, na, pos, nb)
    h = ops.linear_rows_bf16(z0, pk[0], C, bs[0], relu_cols=C)
    h = ops.linear_rows_bf16(h, pk[1], C, bs[1], relu_cols=C)
    me0 = ops.linear_rows_bf16(h, pk[2], C, bs[2])
    qn0 = ops.linear_rows_bf16(yp0, pk[3], C, bs[3])
    return y0, me0, qn0
for name, fn in (('fused', fused), ('chain', chain)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20): out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): gr.replay()
    e1.record(); torch.cuda.synchronize()
    print(name, 'us per call: %.2f' % (e0.elapsed_time(e1) * 1e3 / 400))
