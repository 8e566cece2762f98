import sys, torch
sys.path.insert(0, '/root/repo')
import cgg_amd
from cgg_amd import ops
from cgg_amd.query_decoder import _XAttnFn, pack_bool_mask
dev = torch.device('cuda')
B, Q, H, E = 16, 100, 8, 256
tot = 0
for S, reps in ((1024, 3), (4096, 3), (16384, 3), (100, 9)):
    q = torch.randn(B, Q, E, device=dev, requires_grad=True)
    kv = torch.randn(B, S, 2 * E, device=dev, requires_grad=True)
    bits = pack_bool_mask(torch.rand(B, Q, S) < 0.5).to(dev) if S != 100 else None
    if bits is not None: ops.attn_mask_fix_full_rows(bits, S)
    go = torch.randn(B, Q, E, device=dev)
    def f():
        out = _XAttnFn.apply(q, kv, bits, H)
        out.backward(go)
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print('S=%5d fwd+bwd %.2f ms  (x%d per step)' % (S, ms, reps)); tot += ms * reps
print('attention fwd+bwd per training step ~ %.1f ms' % tot)
