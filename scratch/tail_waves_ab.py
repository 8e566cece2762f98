"""cgg_encoder_layer_tail_x3 at configs[1] / configs[2] row counts: CGG_TAIL_WAVES=4 (round 3-5) vs 8 (round 6: two wavefronts per SIMD).
Run once per setting."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cgg_amd  # noqa: F401
from cgg_amd import ops
dev = torch.device('cuda')
C, F = 256, 1024
g = torch.Generator().manual_seed(0)
r = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k).to(dev)
for M in (43008, 344064):
    a, x, pos = r(M, C), r(M, C), r(21504, C)
    wo, bo, w1, b1, w2, b2 = r(C, C, k=1 / 16), r(C), r(F, C, k=1 / 16), r(F), r(C, F, k=1 / 32), r(C)
    n0, n1 = (r(C), r(C), 1e-5), (r(C), r(C), 1e-5)
    pk = [ops.pack_linear_weight_x3(w) for w in (wo, w1, w2)]
    xe = ops.x3a_encode(x)
    for name, fn in (('x3a rows, y only', lambda: ops.encoder_layer_tail_x3(a, xe, pk[0], bo, n0, pk[1], b1, pk[2], b2, n1, x3a=True)),
                     ('x3a rows, y + pos too', lambda: ops.encoder_layer_tail_x3(a, xe, pk[0], bo, n0, pk[1], b1, pk[2], b2, n1, pos=pos, want_pos=True, x3a=True)),
                     ('f32 rows', lambda: ops.encoder_layer_tail_x3(a, x, pk[0], bo, n0, pk[1], b1, pk[2], b2, n1))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            fn()
        e.record(); torch.cuda.synchronize()
        t = s.elapsed_time(e) / 20 * 1e3
        fl = 2.0 * M * (C * C + 2 * C * F)
        print(f'CGG_TAIL_WAVES={os.environ.get("CGG_TAIL_WAVES", "8 (default)")} M={M} {name}: {t:.1f} us = {fl / t / 1e6:.0f} TF/s = {fl / t / 1e6 / 833.3:.3f} of 833 TF')
