"""cgg_encoder_layer_tail_x3 at configs[1] (M = 43 008 rows, F = 1024) vs the three x3 GEMMs + two LayerNorm passes it replaces."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cgg_amd
from cgg_amd import ops
dev = torch.device('cuda')
M, C, F = 43008, 256, 1024
g = torch.Generator().manual_seed(0)
r = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k).to(dev)
a, x, pos = r(M, C), r(M, C), r(M // 2, C)
wo, bo, w1, b1, w2, b2 = r(C, C, k=1 / 16), r(C), r(F, C, k=1 / 16), r(F), r(C, F, k=1 / 32), r(C)
n0, n1 = (r(C), r(C), 1e-5), (r(C), r(C), 1e-5)
pk = [ops.pack_linear_weight_x3(w) for w in (wo, w1, w2)]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
def fused(): return ops.encoder_layer_tail_x3(a, x, pk[0], bo, n0, pk[1], b1, pk[2], b2, n1, pos=pos, want_pos=True)
def chain():
    y = ops.gemm_x3(a, pk[0], C, bo, res=x)
    x1 = ops.add_layernorm_stream(y, None, n0[0], n0[1], 1e-5, want_bf16=False)[0]
    h = ops.gemm_x3(x1, pk[1], F, b1, relu=True)
    y = ops.gemm_x3(h, pk[2], C, b2, res=x1)
    return ops.add_layernorm_stream(y, None, n1[0], n1[1], 1e-5, want_bf16=False)[0]
xe = ops.x3a_encode(x)
pv = [pk[0], ops.pack_tail_v2_weight_x3(w1), ops.pack_tail_v2_weight_x3(w2)]
def fused_x3a(): return ops.encoder_layer_tail_x3(a, xe, pk[0], bo, n0, pk[1], b1, pk[2], b2, n1, pos=pos, want_pos=True, x3a=True)
def fused_v2(): return ops.encoder_layer_tail_x3(a, xe, pv[0], bo, n0, pv[1], b1, pv[2], b2, n1, pos=pos, want_pos=True, x3a=True, v2=True)
tf, tc = timeit(fused), timeit(chain)
ta, tv = timeit(fused_x3a), timeit(fused_v2)
fl = 2.0 * M * (C * C + 2 * C * F)
print('fused %.1f us (%.1f TF eff = %.3f of the f16 x 3 peak 833 TF), chain %.1f us; max |diff| %.2e' % (
    tf, fl / tf / 1e6, fl / tf / 1e6 / 833.3, tc, (fused()[0] - chain()).abs().max().item()))
print('x3a rows: LDS-image kernel %.1f us (%.3f), register-chained v2 %.1f us (%.1f TF eff = %.3f of the peak); max |v2 - v1| %.2e' % (
    ta, fl / ta / 1e6 / 833.3, tv, fl / tv / 1e6, fl / tv / 1e6 / 833.3,
    (ops.x3a_decode(fused_v2()[0]) - ops.x3a_decode(fused_x3a()[0])).abs().max().item()))
