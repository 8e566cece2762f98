"""Round 4: cgg_gemm_x3s / cgg_conv_x3s_nhwc (x3a rows, LDS-DMA) vs round 3's cgg_gemm_x3 / cgg_conv_x3_nhwc at the shapes of
parity mode's step (configs[1]: R50, 1024^2, batch 2), every tile configuration. argv: [configs csv | 'auto'] [filter substring]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cgg_amd
from cgg_amd import ops
from cgg_amd._lib import load
dev = torch.device('cuda')
lib = load()
CFGS = [int(c) for c in sys.argv[1].split(',')] if len(sys.argv) > 1 and sys.argv[1] != 'auto' else list(range(18))
AUTO_ONLY = len(sys.argv) > 1 and sys.argv[1] == 'auto'
FILT = sys.argv[2] if len(sys.argv) > 2 else ''
BM = [256, 256, 128, 128, 128, 64, 128, 64, 256, 256, 128, 64, 64, 64, 128, 64, 64, 64]
BN = [256, 128, 256, 128, 128, 128, 64, 64, 64, 256, 128, 128, 64, 64, 64, 128, 64, 128]
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
tot = {'old': 0.0, 'best': 0.0, 'auto': 0.0, 'fl': 0.0}
def report(tag, count, fl, t_old, ts, t_auto):
    best = min(ts, key=lambda c: ts[c]) if ts else -1
    tb = ts[best] if ts else float('nan')
    tot['old'] += t_old * count; tot['best'] += tb * count; tot['auto'] += t_auto * count; tot['fl'] += fl * count
    cols = ' '.join(f'{c}:{ts[c]:6.1f}' for c in sorted(ts))
    print(f'{tag:40s} x{count} old {t_old:6.1f} ({fl / t_old / 1e6:5.0f} TF) auto {t_auto:6.1f} ({fl / t_auto / 1e6:5.0f} TF) best cfg {best} {tb:6.1f} ({fl / tb / 1e6:5.0f} TF) | {cols}', flush=True)
def run_cfgs(fn, M, N):
    # round 5: the tile configuration is a call argument (`cfg=`), no process-global override
    ts = {}
    if not AUTO_ONLY:
        for c in CFGS:
            if BN[c % 100] >= 2 * N and BN[c % 100] > 64: continue       # tile far wider than the problem
            ts[c] = timeit(lambda: fn(c))
    return ts, timeit(lambda: fn(-1))
def gemm(M, N, K, count=1, tag='gemm', res=False):
    name = f'{tag} {M}x{N}x{K}'
    if FILT and FILT not in name: return
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / K**0.5; b = torch.randn(N, device=dev)
    pk = ops.pack_linear_weight_x3(w); xe = ops.x3a_encode(x)
    y = torch.empty(M, N, device=dev)
    t_old = timeit(lambda: ops.gemm_x3(x, pk, N, b, out=y))
    ts, ta = run_cfgs(lambda c: ops.gemm_x3s(xe, pk, N, b, out=y, out_split=True, cfg=c), M, N)
    report(name, count, 2.0 * M * N * K, t_old, ts, ta)
def conv(B, H, C, N, k, s, count=1, res=False):
    name = f'conv {B}x{H}x{H}x{C} -> {N} k{k} s{s}' + (' +res' if res else '')
    if FILT and FILT not in name: return
    x = torch.randn(B, H, H, C, device=dev); w = torch.randn(N, C, k, k, device=dev) / (C * k * k)**0.5; b = torch.randn(N, device=dev)
    pk = ops.pack_conv_weight_x3(w); xe = ops.x3a_encode(x)
    OH = (H + 2 * (k // 2) - k) // s + 1
    r = torch.randn(B, OH, OH, N, device=dev) if res else None
    re_ = ops.x3a_encode(r) if res else None
    t_old = timeit(lambda: ops.conv_x3_nhwc(x, pk, N, k, s, k // 2, b, res=r, relu=True))
    ts, ta = run_cfgs(lambda c: ops.conv_x3s_nhwc(xe, pk, N, k, s, k // 2, b, res=re_, relu=True, cfg=c), B * OH * OH, N)
    report(name, count, 2.0 * B * OH * OH * N * C * k * k, t_old, ts, ta)
# ---- ResNet-50 at 1024^2, batch 2 (after the stem: 256^2 x 64) ----
conv(2, 256, 64, 256, 1, 1, 1); conv(2, 256, 64, 256, 1, 1, 3, res=True); conv(2, 256, 64, 64, 1, 1, 1); conv(2, 256, 64, 64, 3, 1, 3); conv(2, 256, 256, 64, 1, 1, 2)
conv(2, 256, 256, 512, 1, 2, 1); conv(2, 256, 256, 128, 1, 1, 1); conv(2, 256, 128, 128, 3, 2, 1); conv(2, 128, 128, 512, 1, 1, 4, res=True)
conv(2, 128, 512, 128, 1, 1, 3); conv(2, 128, 128, 128, 3, 1, 3)
conv(2, 128, 512, 1024, 1, 2, 1); conv(2, 128, 512, 256, 1, 1, 1); conv(2, 128, 256, 256, 3, 2, 1); conv(2, 64, 256, 1024, 1, 1, 6, res=True)
conv(2, 64, 1024, 256, 1, 1, 5); conv(2, 64, 256, 256, 3, 1, 5)
conv(2, 64, 1024, 2048, 1, 2, 1); conv(2, 64, 1024, 512, 1, 1, 1); conv(2, 64, 512, 512, 3, 2, 1); conv(2, 32, 512, 2048, 1, 1, 3, res=True)
conv(2, 32, 2048, 512, 1, 1, 2); conv(2, 32, 512, 512, 3, 1, 2)
print('backbone total: old %.0f us, auto %.0f us, best %.0f us; %.1f GF -> %.1f / %.1f / %.1f TF' % (tot['old'], tot['auto'], tot['best'], tot['fl'] / 1e9, tot['fl'] / tot['old'] / 1e6, tot['fl'] / tot['auto'] / 1e6, tot['fl'] / tot['best'] / 1e6))
t0 = dict(tot)
# ---- pixel decoder convs + encoder + K/V ----
gemm(2048, 256, 2048, 1, 'input conv'); gemm(8192, 256, 1024, 1, 'input conv'); gemm(32768, 256, 512, 1, 'input conv')
gemm(131072, 256, 256, 2, 'lateral / mask_feature'); conv(2, 256, 256, 256, 3, 1, 1)
gemm(43008, 256, 256, 6, 'value proj'); gemm(43008, 288, 256, 6, 'offsets')
gemm(16384, 512, 256, 3, 'kv proj'); gemm(4096, 512, 256, 3, 'kv proj'); gemm(1024, 512, 256, 3, 'kv proj')
gemm(16384, 1536, 256, 1, 'kv proj x3 layers'); gemm(4096, 1536, 256, 1, 'kv proj x3 layers'); gemm(1024, 1536, 256, 1, 'kv proj x3 layers')
d = {k: tot[k] - t0[k] for k in tot}
print('pixel decoder + K/V total: old %.0f us, auto %.0f us, best %.0f us; %.1f GF -> %.1f / %.1f / %.1f TF' % (d['old'], d['auto'], d['best'], d['fl'] / 1e9, d['fl'] / d['old'] / 1e6, d['fl'] / d['auto'] / 1e6, d['fl'] / d['best'] / 1e6))
print('ALL: old %.0f us, auto %.0f us, best %.0f us; %.1f / %.1f / %.1f TF' % (tot['old'], tot['auto'], tot['best'], tot['fl'] / tot['old'] / 1e6, tot['fl'] / tot['auto'] / 1e6, tot['fl'] / tot['best'] / 1e6))
