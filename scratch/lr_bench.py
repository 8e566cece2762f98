import sys, torch
sys.path.insert(0, '.')
import cgg_amd
from cgg_amd import ops
dev = 'cuda'
def bench(f, n=200):
    for _ in range(10): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (M, K, N) in [(200, 256, 256), (200, 256, 2048), (200, 2048, 256), (200, 256, 768), (32, 256, 32), (128, 256, 32), (200, 256, 32)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    for split in (False, True):
        t = bench(lambda: ops.linear_rows(x, w, b, split=split))
        print((M, K, N), 'split' if split else 'bf16', '%.2f us' % t, flush=True)
a = torch.randn(200, 256, device=dev); gm = torch.ones(256, device=dev); bt = torch.zeros(256, device=dev)
print('add_layernorm', '%.2f us' % bench(lambda: ops.add_layernorm(a, None, gm, bt)))
q = torch.randn(2, 100, 256, device=dev)
for S in (100, 1024, 4096, 16384):
    kv = torch.randn(2, S, 512, device=dev)
    print('xattn S=%d' % S, '%.2f us' % bench(lambda: ops.masked_xattn(q, kv, None, 8), 50))
