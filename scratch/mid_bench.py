import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from cgg_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(70)
M, C, Q = 200, 256, 100
core = torch.randn(M, C, generator=g).to(dev)
res = torch.randn(M, C, generator=g).to(dev)
pos = torch.randn(Q, C, generator=g).to(dev)
norm = (torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev), 1e-5)
wo, bo = (torch.randn(C, C, generator=g) / 16).to(dev), torch.randn(C, generator=g).to(dev)
wqkv, bqkv = (torch.randn(3 * C, C, generator=g) / 16).to(dev), torch.randn(3 * C, generator=g).to(dev)
pwo, pqkv = ops.pack_linear_weight(wo), ops.pack_linear_weight(wqkv)
def fused(): return ops.decoder_mid(core, pwo, bo, res, norm, pos, (pqkv, bqkv))
def fused1(): return ops.decoder_mid(core, pwo, bo, res, norm)
def chain():
    x1_0, x1p_0 = ops.linear_rows_bf16(core, pwo, C, bo, res=res, ln=norm, pos=pos, want_pos=True)
    return ops.linear_rows_bf16_qkv(x1p_0, x1_0, pqkv, bqkv, C)
def chain1(): return ops.linear_rows_bf16(core, pwo, C, bo, res=res, ln=norm)
for name, fn in (('mid+qkv', fused), ('chain+qkv', chain), ('mid', fused1), ('chain', chain1)):
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
