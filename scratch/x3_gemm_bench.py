"""cgg_gemm_x3 / cgg_conv_x3_nhwc at the shapes of parity mode's step (configs[1]: R50, 1024^2, batch 2) vs the f32 library
calls they replace. argv[1] = 'lib' also times the f32 library."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cgg_amd
from cgg_amd import ops
dev = torch.device('cuda')
LIB = len(sys.argv) > 1 and sys.argv[1] == 'lib'
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
print('%-44s %5s %9s %9s %9s %9s' % ('shape', 'count', 'x3 us', 'TF eff', 'f32lib us', 'TF'))
tot = [0.0, 0.0]
def gemm(M, N, K, count=1, tag='gemm'):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K**0.5; b = torch.randn(N, device=dev)
    pk = ops.pack_linear_weight_x3(w)
    t = timeit(lambda: ops.gemm_x3(x, pk, N, b))
    t2 = timeit(lambda: torch.nn.functional.linear(x, w, b)) if LIB else float('nan')
    fl = 2.0 * M * N * K
    tot[0] += t * count; tot[1] += fl * count
    print('%-44s %5d %9.1f %9.1f %9.1f %9.1f' % (f'{tag} {M}x{N}x{K}', count, t, fl / t / 1e6, t2, fl / t2 / 1e6))
def conv(B, H, C, N, k, s, count=1):
    x = torch.randn(B, H, H, C, device=dev); w = torch.randn(N, C, k, k, device=dev) / (C * k * k)**0.5; b = torch.randn(N, device=dev)
    pk = ops.pack_conv_weight_x3(w)
    t = timeit(lambda: ops.conv_x3_nhwc(x, pk, N, k, s, k // 2, b))
    t2 = float('nan')
    if LIB:
        xn = x.permute(0, 3, 1, 2).contiguous()
        t2 = timeit(lambda: torch.nn.functional.conv2d(xn, w, b, stride=s, padding=k // 2))
    OH = (H + 2 * (k // 2) - k) // s + 1
    fl = 2.0 * B * OH * OH * N * C * k * k
    tot[0] += t * count; tot[1] += fl * count
    print('%-44s %5d %9.1f %9.1f %9.1f %9.1f' % (f'conv {B}x{H}x{H}x{C} -> {N} k{k} s{s}', count, t, fl / t / 1e6, t2, fl / t2 / 1e6))
# ---- ResNet-50 at 1024^2, batch 2 (after the stem: 256^2 x 64) ----
conv(2, 256, 64, 256, 1, 1, 4); conv(2, 256, 64, 64, 1, 1, 1); conv(2, 256, 64, 64, 3, 1, 3); conv(2, 256, 256, 64, 1, 1, 2)
conv(2, 256, 256, 512, 1, 2, 1); conv(2, 256, 256, 128, 1, 1, 1); conv(2, 256, 128, 128, 3, 2, 1); conv(2, 128, 128, 512, 1, 1, 4)
conv(2, 128, 512, 128, 1, 1, 3); conv(2, 128, 128, 128, 3, 1, 3)
conv(2, 128, 512, 1024, 1, 2, 1); conv(2, 128, 512, 256, 1, 1, 1); conv(2, 128, 256, 256, 3, 2, 1); conv(2, 64, 256, 1024, 1, 1, 6)
conv(2, 64, 1024, 256, 1, 1, 5); conv(2, 64, 256, 256, 3, 1, 5)
conv(2, 64, 1024, 2048, 1, 2, 1); conv(2, 64, 1024, 512, 1, 1, 1); conv(2, 64, 512, 512, 3, 2, 1); conv(2, 32, 512, 2048, 1, 1, 3)
conv(2, 32, 2048, 512, 1, 1, 2); conv(2, 32, 512, 512, 3, 1, 2)
print('backbone total: %.0f us, %.1f GF, %.1f TF eff' % (tot[0], tot[1] / 1e9, tot[1] / tot[0] / 1e6))
t0 = list(tot)
# ---- pixel decoder convs + encoder + K/V ----
gemm(2048, 256, 2048, 1, 'input conv'); gemm(8192, 256, 1024, 1, 'input conv'); gemm(32768, 256, 512, 1, 'input conv')
gemm(131072, 256, 256, 2, 'lateral / mask_feature'); conv(2, 256, 256, 256, 3, 1, 1)
gemm(43008, 256, 256, 12, 'value / out proj'); gemm(43008, 288, 256, 6, 'offsets'); gemm(43008, 1024, 256, 6, 'ffn1'); gemm(43008, 256, 1024, 6, 'ffn2')
gemm(16384, 512, 256, 6, 'kv proj'); gemm(4096, 512, 256, 6, 'kv proj'); gemm(1024, 512, 256, 6, 'kv proj')
print('pixel decoder + K/V total: %.0f us, %.1f GF, %.1f TF eff' % (tot[0] - t0[0], (tot[1] - t0[1]) / 1e9, (tot[1] - t0[1]) / (tot[0] - t0[0]) / 1e6))
