"""tile-configuration sweep of cgg_gemm_x3s at one shape: python scratch/x3s_cfg_sweep.py M N K [res_mod]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cgg_amd  # noqa: F401
from cgg_amd import ops
dev = torch.device('cuda')
M, N, K = (int(v) for v in sys.argv[1:4])
res_mod = int(sys.argv[4]) if len(sys.argv) > 4 else 0
g = torch.Generator().manual_seed(0)
a = ops.x3a_encode(torch.randn(M, K, generator=g).to(dev))
w = (torch.randn(N, K, generator=g) * 0.05).to(dev)
b = torch.randn(N, generator=g).to(dev)
pk = ops.pack_linear_weight_x3(w)
res = torch.randn(res_mod, N, generator=g).to(dev) if res_mod else None
out = torch.empty(M, N, device=dev)
# rotate through distinct A buffers so that a repeated launch does not find its operands in L2 / the Infinity Cache
abufs = [a] + [a.clone() for _ in range(7)]
for cfg in [-1] + list(range(18)):
    try:
        def run(i):
            ops.gemm_x3s(abufs[i % 8], pk, N, b, res=res, res_mod=res_mod, out=out, cfg=cfg)
        for i in range(4):
            run(i)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(24):
            run(i)
        e.record(); torch.cuda.synchronize()
        t = s.elapsed_time(e) / 24 * 1e3
        print(f'M={M} N={N} K={K} res_mod={res_mod} cfg {cfg:2d}: {t:7.1f} us  {2.0 * M * N * K / t / 1e6 / 833.3:.3f} of 833 TF')
    except Exception as ex:
        print(f'cfg {cfg}: {str(ex)[:80]}')
