"""Two launches-only workloads for PMC passes of cgg_gemm_x3_kernel: argv[1] = 'conv' | 'gemm' | 'gemm256'."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cgg_amd
from cgg_amd import ops
dev = torch.device('cuda')
which = sys.argv[1] if len(sys.argv) > 1 else 'conv'
if which == 'conv':
    x = torch.randn(2, 256, 256, 256, device=dev); w = torch.randn(256, 256, 3, 3, device=dev) / 48; b = torch.randn(256, device=dev)
    pk = ops.pack_conv_weight_x3(w)
    fn = lambda: ops.conv_x3_nhwc(x, pk, 256, 3, 1, 1, b)
else:
    M, N, K = (43008, 1024, 256) if which == 'gemm' else (43008, 256, 256)
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / 16; b = torch.randn(N, device=dev)
    pk = ops.pack_linear_weight_x3(w)
    fn = lambda: ops.gemm_x3(x, pk, N, b)
for _ in range(5):
    fn()
torch.cuda.synchronize()
