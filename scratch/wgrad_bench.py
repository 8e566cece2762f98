"""cgg_wgrad_x3 vs the split-K f32 library GEMM at configs[2]'s encoder shapes (M = 16 x 21 504 rows)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cgg_amd
from cgg_amd import ops
dev = torch.device('cuda')
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
M = 344064
for N, K in ((256, 256), (288, 256), (1024, 256), (256, 1024)):
    dy = torch.randn(M, N, device=dev) * 0.1; x = torch.randn(M, K, device=dev)
    S = next((s for s in (32, 16, 8, 4, 2) if M % s == 0 and M // s >= 4096), 1)
    t_lib = timeit(lambda: torch.bmm(dy.view(S, M // S, N).transpose(1, 2), x.view(S, M // S, K)).sum(0))
    t_x3 = timeit(lambda: ops.wgrad_x3(dy, x))
    fl = 2.0 * M * N * K
    print(f'wgrad {M}x{N}x{K}: library split-K {t_lib:8.1f} us ({fl / t_lib / 1e6:5.0f} TF)   x3 {t_x3:8.1f} us ({fl / t_x3 / 1e6:5.0f} TF)', flush=True)
