"""Launches of cgg_gemm_x3 at the configs[2] encoder shapes only (for the PMC passes of scratch/x3_train_gemm_pmc.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cgg_amd
from cgg_amd import ops
M = 344064
for K, N in [(256, 256), (256, 1024), (1024, 256)]:
    a = torch.randn(M, K, device='cuda')
    pk = ops.pack_linear_weight_x3(torch.randn(N, K, device='cuda') * 0.05)
    out = torch.empty(M, N, device='cuda')
    for _ in range(3):
        ops.gemm_x3(a, pk, N, out=out)
    torch.cuda.synchronize()
