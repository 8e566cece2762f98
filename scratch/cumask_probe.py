import sys, os, torch, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cgg_amd import ops
dev = torch.device('cuda:0')
x = torch.randn(4096, 4096, device=dev)
def bench(stream, n=20):
    with torch.cuda.stream(stream):
        for _ in range(3): y = x @ x
        stream.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): y = x @ x
        stream.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print('default stream   : %.3f ms per 4096^3 f32 matmul' % bench(torch.cuda.Stream(dev)))
for share in (1, 2, 3):
    bits = [1 if ((c >> 3) & 3) < share else 0 for c in range(256)]
    s = ops.masked_stream(dev, bits)
    print('masked %d/4 CUs   : %.3f ms' % (share, bench(s)), flush=True)
print('ok')
