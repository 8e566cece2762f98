"""attention forward(+lse) / backward at configs[2] shapes: HIP kernels vs the torch formulation they replace."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgg_amd
from cgg_amd import ops
from cgg_amd.query_decoder import pack_bool_mask, xattn_backward_torch
dev = torch.device('cuda')
B, Q, H, E = 16, 100, 8, 256


def timeit(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tot = dict(fwd=0, bwd=0, torch=0)
for S, reps in ((1024, 3), (4096, 3), (16384, 3), (100, 9)):
    q = torch.randn(B, Q, E, device=dev)
    kv = torch.randn(B, S, 2 * E, device=dev)
    bits = pack_bool_mask(torch.rand(B, Q, S) < 0.5).to(dev) if S != 100 else None
    if bits is not None: ops.attn_mask_fix_full_rows(bits, S)
    go = torch.randn(B, Q, E, device=dev)
    out, lse = ops.masked_xattn(q, kv, bits, H, return_lse=True)
    fwd = timeit(lambda: ops.masked_xattn(q, kv, bits, H, return_lse=True))
    bwd = timeit(lambda: ops.masked_xattn_backward(q, kv, bits, out, lse, go, H))
    tch = timeit(lambda: xattn_backward_torch(q, kv, bits, go, H))
    flops = 5 * 2.0 * B * H * Q * S * 32
    print('S=%5d fwd %.3f ms  bwd %.3f ms (%.1f TFLOP/s f32 MFMA)  torch-op bwd %.3f ms   (x%d per step)' % (S, fwd, bwd, flops / bwd / 1e9, tch, reps))
    tot['fwd'] += fwd * reps; tot['bwd'] += bwd * reps; tot['torch'] += tch * reps
print('per training step: fwd %.1f ms, HIP bwd %.1f ms, torch-op bwd %.1f ms' % (tot['fwd'], tot['bwd'], tot['torch']))
