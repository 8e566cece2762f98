import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cgg_amd
from cgg_amd import ops
from cgg_amd._lib import load
dev = torch.device('cuda'); lib = load()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
def conv(B, H, C, N, k, s, cfgs):
    x = torch.randn(B, H, H, C, device=dev); w = torch.randn(N, C, k, k, device=dev) / (C * k * k)**0.5; b = torch.randn(N, device=dev)
    pk = ops.pack_conv_weight_x3(w); xe = ops.x3a_encode(x)
    for c in cfgs:
        r = []
        for a in (0, 1, 2):
            lib.cgg_gemm_x3s_force_config(c + 100 * a)
            r.append(timeit(lambda: ops.conv_x3s_nhwc(xe, pk, N, k, s, k // 2, b, relu=True)))
        print(f'conv {B}x{H}x{H}x{C}->{N} k{k}s{s} cfg {c}: full {r[0]:.1f}  noDMA {r[1]:.1f}  noMFMA {r[2]:.1f}', flush=True)
    lib.cgg_gemm_x3s_force_config(-1)
conv(2, 256, 256, 256, 3, 1, [0, 9, 16, 1, 2, 4])
conv(2, 64, 256, 256, 3, 1, [3, 5, 7])
conv(2, 32, 512, 512, 3, 1, [5, 7])
conv(2, 256, 64, 256, 1, 1, [0, 4])
conv(2, 256, 64, 64, 3, 1, [7, 8])
