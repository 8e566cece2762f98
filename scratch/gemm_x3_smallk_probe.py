"""gemm_x3 at the small K / N of a narrow ResNet stage (K = 32: a single chunk), plain and amax-scaled, against float64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cgg_amd
from cgg_amd import ops
dev = torch.device('cuda')
torch.manual_seed(0)
for M, N, K in ((32768, 128, 32), (32768, 32, 128), (32768, 128, 64), (8192, 128, 32), (32768, 32, 32), (32768, 64, 32), (32768, 256, 32)):
    w = torch.randn(N, K, device=dev) * 0.2
    for scale in (1.0, 1e-5):
        a = torch.randn(M, K, device=dev) * scale
        pk = ops.pack_linear_weight_x3(w)
        want = a.double() @ w.double().t()
        for name, y in (('fixed', ops.gemm_x3(a, pk, N)), ('amax', ops.gemm_x3(a, pk, N, amax=ops.absmax(a)))):
            err = (y.double() - want).abs().max().item() / want.abs().max().item()
            print(f'M={M} N={N} K={K} |a|~{scale:g} {name}: rel err {err:.2e}', flush=True)
