import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cgg_amd
from cgg_amd import ops
from cgg_amd._lib import load
dev = torch.device('cuda'); lib = load()
def timeit(fns, n=24):
    for f in fns: f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fns[i % len(fns)]()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
def conv(B, H, C, N, k, s, res, cfgs, pool=4):
    w = torch.randn(N, C, k, k, device=dev) / (C * k * k)**0.5; b = torch.randn(N, device=dev)
    pk = ops.pack_conv_weight_x3(w)
    OH = (H + 2 * (k // 2) - k) // s + 1
    xs = [torch.randn(B, H, H, C, device=dev) for _ in range(pool)]
    xe = [ops.x3a_encode(x) for x in xs]
    rs = [torch.randn(B, OH, OH, N, device=dev) for _ in range(pool)] if res else [None] * pool
    re_ = [ops.x3a_encode(r) for r in rs] if res else [None] * pool
    t_old = timeit([lambda i=i: ops.conv_x3_nhwc(xs[i], pk, N, k, s, k // 2, b, res=rs[i], relu=True) for i in range(pool)])
    line = f'[EPIMAP={os.environ.get("CGG_XS_EPIMAP","0")}] conv {B}x{H}x{H}x{C}->{N} k{k}s{s}{"+res" if res else ""}: old {t_old:.1f} |'
    for c in cfgs:
        lib.cgg_gemm_x3s_force_config(c)
        t = timeit([lambda i=i: ops.conv_x3s_nhwc(xe[i], pk, N, k, s, k // 2, b, res=re_[i], relu=True) for i in range(pool)])
        line += f' {c}: {t:.1f}'
        if res:
            t = timeit([lambda i=i: ops.conv_x3s_nhwc(xe[i], pk, N, k, s, k // 2, b, res=rs[i], res_split=False, relu=True) for i in range(pool)])
            line += f' (f32res {t:.1f})'
            t = timeit([lambda i=i: ops.conv_x3s_nhwc(xe[i], pk, N, k, s, k // 2, b, res=rs[i], res_split=False, relu=True, out_split=False) for i in range(pool)])
            line += f' (f32res+f32out {t:.1f})'
    lib.cgg_gemm_x3s_force_config(-1)
    print(line, flush=True)
conv(2, 256, 64, 256, 1, 1, True, [14, 13, 4, 15])
conv(2, 256, 64, 256, 1, 1, False, [14, 13, 4, 15])
conv(2, 128, 128, 512, 1, 1, True, [14, 13, 4, 15])
conv(2, 64, 256, 1024, 1, 1, True, [4, 14, 2, 15])
