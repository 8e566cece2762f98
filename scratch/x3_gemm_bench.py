"""cgg_gemm_x3 / cgg_conv_x3_nhwc at the shapes of parity mode's step (configs[1]) vs the f32 library calls they replace."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cgg_amd
from cgg_amd import ops
dev = torch.device('cuda')
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
print('%-44s %9s %9s %9s %9s' % ('shape', 'x3 us', 'TF eff', 'f32lib us', 'TF'))
for M, N, K in [(43008, 256, 256), (43008, 288, 256), (43008, 1024, 256), (43008, 256, 1024), (131072, 256, 256), (131072, 64, 256),
                (131072, 256, 64), (32768, 512, 128), (8192, 1024, 256), (2048, 2048, 512), (32768, 768, 256), (8192, 768, 256)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K**0.5; b = torch.randn(N, device=dev)
    pk = ops.pack_linear_weight_x3(w)
    t = timeit(lambda: ops.gemm_x3(x, pk, N, b))
    t2 = timeit(lambda: torch.nn.functional.linear(x, w, b))
    fl = 2.0 * M * N * K
    print('%-44s %9.1f %9.1f %9.1f %9.1f' % (f'gemm {M}x{N}x{K}', t, fl / t / 1e6, t2, fl / t2 / 1e6))
for B, H, C, N, k, s in [(2, 256, 64, 64, 3, 1), (2, 128, 128, 128, 3, 1), (2, 64, 256, 256, 3, 1), (2, 32, 512, 512, 3, 1),
                         (2, 256, 256, 256, 3, 1), (2, 256, 128, 128, 3, 2)]:
    x = torch.randn(B, H, H, C, device=dev); w = torch.randn(N, C, k, k, device=dev) / (C * k * k)**0.5; b = torch.randn(N, device=dev)
    pk = ops.pack_conv_weight_x3(w)
    t = timeit(lambda: ops.conv_x3_nhwc(x, pk, N, k, s, k // 2, b))
    xn = x.permute(0, 3, 1, 2).contiguous()
    t2 = timeit(lambda: torch.nn.functional.conv2d(xn, w, b, stride=s, padding=k // 2))
    OH = (H + 2 * (k // 2) - k) // s + 1
    fl = 2.0 * B * OH * OH * N * C * k * k
    print('%-44s %9.1f %9.1f %9.1f %9.1f' % (f'conv {B}x{H}x{H}x{C} -> {N} k{k} s{s}', t, fl / t / 1e6, t2, fl / t2 / 1e6))
